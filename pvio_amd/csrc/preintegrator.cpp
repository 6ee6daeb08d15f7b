// preintegrator.cpp -- host-side IMU pre-integration (serial recurrence; SURVEY.md 8a row A4 keeps it on the host).
//
// Replaces PreIntegrator::{reset, increment, integrate, compute_sqrt_inv_cov}
// (pvio/src/pvio/estimation/preintegrator.cpp:24-100).  Called by the adapter for every consecutive frame pair
// at solve entry, like bundle_adjustor.cpp:224 does.
#include <cstring>
#include <vector>

#include "../../include/pvio_hip.h"
#include "pv_math.h"

namespace {

using namespace pv;

struct State {
    double t = 0, q[4] = {0, 0, 0, 1}, p[3] = {0, 0, 0}, v[3] = {0, 0, 0};
    double cov[15 * 15];
    double dq_dbg[9], dp_dbg[9], dp_dba[9], dv_dbg[9], dv_dba[9];
};

void m3_axpy(double *o, double s, const double *a) {
    for (int k = 0; k < 9; ++k) o[k] += s * a[k];
}

// one Euler step with the sample held over dt (preintegrator.cpp:39-82)
void increment(State &s, double dt, const double *gyr, const double *acc, const double *bg, const double *ba, const pvio_imu_noise &nz) {
    double w[3], a[3], wdt[3];
    v3_sub(w, gyr, bg);
    v3_sub(a, acc, ba);
    v3_set(wdt, w[0] * dt, w[1] * dt, w[2] * dt);
    double Rdq[9], e[4], ec[4], RexpT[9], Jr[9], Ra[9];
    q_to_mat(Rdq, s.q);
    q_expmap(e, wdt);
    q_conj(ec, e);
    q_to_mat(RexpT, ec);
    so3_right_jacobian(Jr, wdt);
    m3_mul_hat(Ra, Rdq, a); // R(dq) [a]x
    // covariance: Sigma9 <- A Sigma9 A^T + B diag(cov_w/dt', cov_a/dt') B^T ; bias random walks (:44-65)
    {
        double A[81] = {0}, B[54] = {0}, Wn[36] = {0};
        for (int i = 0; i < 9; ++i) A[i * 9 + i] = 1.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                A[(0 + i) * 9 + j] = RexpT[3 * i + j];
                A[(6 + i) * 9 + j] = -dt * Ra[3 * i + j];
                A[(3 + i) * 9 + j] = -0.5 * dt * dt * Ra[3 * i + j];
                A[(3 + i) * 9 + 6 + j] = (i == j) ? dt : 0.0;
                B[(0 + i) * 6 + j] = dt * Jr[3 * i + j];
                B[(6 + i) * 6 + 3 + j] = dt * Rdq[3 * i + j];
                B[(3 + i) * 6 + 3 + j] = 0.5 * dt * dt * Rdq[3 * i + j];
            }
        const double inv_dt = 1.0 / (dt > 1.0e-7 ? dt : 1.0e-7);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                Wn[i * 6 + j] = nz.cov_w[3 * i + j] * inv_dt;
                Wn[(3 + i) * 6 + 3 + j] = nz.cov_a[3 * i + j] * inv_dt;
            }
        double AC[81], BW[54], T[81];
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) {
                double acc2 = 0;
                for (int k = 0; k < 9; ++k) acc2 += A[i * 9 + k] * s.cov[k * 15 + j];
                AC[i * 9 + j] = acc2;
            }
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 6; ++j) {
                double acc2 = 0;
                for (int k = 0; k < 6; ++k) acc2 += B[i * 6 + k] * Wn[k * 6 + j];
                BW[i * 6 + j] = acc2;
            }
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) {
                double acc2 = 0;
                for (int k = 0; k < 9; ++k) acc2 += AC[i * 9 + k] * A[j * 9 + k];
                for (int k = 0; k < 6; ++k) acc2 += BW[i * 6 + k] * B[j * 6 + k];
                T[i * 9 + j] = acc2;
            }
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) s.cov[i * 15 + j] = T[i * 9 + j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                s.cov[(9 + i) * 15 + 9 + j] += nz.cov_bg[3 * i + j] * dt;
                s.cov[(12 + i) * 15 + 12 + j] += nz.cov_ba[3 * i + j] * dt;
            }
    }
    // bias Jacobians, statement order of the reference (:69-75): each line sees the OLD values on its right
    double Ra_dq[9], t9[9];
    m3_mul(Ra_dq, Ra, s.dq_dbg);
    m3_axpy(s.dp_dbg, dt, s.dv_dbg);
    m3_axpy(s.dp_dbg, -0.5 * dt * dt, Ra_dq);
    m3_axpy(s.dp_dba, dt, s.dv_dba);
    m3_axpy(s.dp_dba, -0.5 * dt * dt, Rdq);
    m3_axpy(s.dv_dbg, -dt, Ra_dq);
    m3_axpy(s.dv_dba, -dt, Rdq);
    m3_mul(t9, RexpT, s.dq_dbg);
    for (int k = 0; k < 9; ++k) s.dq_dbg[k] = t9[k] - dt * Jr[k];
    // mean (:77-80)
    double Rda[3], qn[4];
    q_rot(Rda, s.q, a);
    s.t += dt;
    for (int k = 0; k < 3; ++k) s.p[k] += dt * s.v[k] + 0.5 * dt * dt * Rda[k];
    for (int k = 0; k < 3; ++k) s.v[k] += dt * Rda[k];
    q_mul(qn, s.q, e);
    q_normalize(qn);
    std::memcpy(s.q, qn, sizeof qn);
}

// inverse by LU with partial pivoting, then lower Cholesky: U = LLT(cov^-1).matrixL()^T (:98-100)
bool sqrt_information(const double *cov, double *U) {
    const int n = 15;
    double A[225], inv[225];
    int piv[15];
    std::memcpy(A, cov, sizeof A);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = A[k * n + k] < 0 ? -A[k * n + k] : A[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double x = A[i * n + k] < 0 ? -A[i * n + k] : A[i * n + k];
            if (x > best) best = x, p = i;
        }
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) {
                const double t = A[k * n + j];
                A[k * n + j] = A[p * n + j], A[p * n + j] = t;
            }
            const int t = piv[k];
            piv[k] = piv[p], piv[p] = t;
        }
        for (int i = k + 1; i < n; ++i) {
            const double f = A[i * n + k] / A[k * n + k];
            A[i * n + k] = f;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
        }
    }
    for (int col = 0; col < n; ++col) {
        double x[15];
        for (int i = 0; i < n; ++i) x[i] = piv[i] == col ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < i; ++k) x[i] -= A[i * n + k] * x[k];
        for (int i = n - 1; i >= 0; --i) {
            for (int k = i + 1; k < n; ++k) x[i] -= A[i * n + k] * x[k];
            x[i] /= A[i * n + i];
        }
        for (int i = 0; i < n; ++i) inv[i * n + col] = x[i];
    }
    // lower Cholesky of inv (reads the lower triangle)
    for (int j = 0; j < n; ++j) {
        double d = inv[j * n + j];
        for (int k = 0; k < j; ++k) d -= inv[j * n + k] * inv[j * n + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        inv[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s2 = inv[i * n + j];
            for (int k = 0; k < j; ++k) s2 -= inv[i * n + k] * inv[j * n + k];
            inv[i * n + j] = s2 / d;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) U[i * n + j] = j >= i ? inv[j * n + i] : 0.0;
    return true;
}

} // namespace

extern "C" int32_t pvio_preintegrate(int32_t n, const double *imu_t, const double *imu_w, const double *imu_a, double t_end,
                                     const double bg[3], const double ba[3], const pvio_imu_noise *noise, double delta[11], double cov[225],
                                     double sqrt_inv_cov[225], double jacobian[45]) {
    if (n <= 0 || !imu_t || !imu_w || !imu_a || !bg || !ba || !noise || !delta || !jacobian) return PVIO_ERR_INVALID_ARGUMENT;
    State s;
    std::memset(s.cov, 0, sizeof s.cov);
    for (double *m : {s.dq_dbg, s.dp_dbg, s.dp_dba, s.dv_dbg, s.dv_dba}) std::memset(m, 0, 9 * sizeof(double));
    for (int i = 0; i + 1 < n; ++i) {
        const double dt = imu_t[i + 1] - imu_t[i];
        if (dt < 0) return PVIO_ERR_INVALID_ARGUMENT; // runtime_assert(dt >= 0) in the reference
        increment(s, dt, imu_w + 3 * i, imu_a + 3 * i, bg, ba, *noise);
    }
    if (t_end - imu_t[n - 1] < 0) return PVIO_ERR_INVALID_ARGUMENT;
    increment(s, t_end - imu_t[n - 1], imu_w + 3 * (n - 1), imu_a + 3 * (n - 1), bg, ba, *noise);
    delta[0] = s.t;
    std::memcpy(delta + 1, s.q, 4 * sizeof(double));
    std::memcpy(delta + 5, s.p, 3 * sizeof(double));
    std::memcpy(delta + 8, s.v, 3 * sizeof(double));
    if (cov) std::memcpy(cov, s.cov, sizeof s.cov);
    std::memcpy(jacobian, s.dq_dbg, 9 * sizeof(double));
    std::memcpy(jacobian + 9, s.dp_dbg, 9 * sizeof(double));
    std::memcpy(jacobian + 18, s.dp_dba, 9 * sizeof(double));
    std::memcpy(jacobian + 27, s.dv_dbg, 9 * sizeof(double));
    std::memcpy(jacobian + 36, s.dv_dba, 9 * sizeof(double));
    if (sqrt_inv_cov && !sqrt_information(s.cov, sqrt_inv_cov)) return PVIO_ERR_INVALID_ARGUMENT;
    return PVIO_OK;
}
