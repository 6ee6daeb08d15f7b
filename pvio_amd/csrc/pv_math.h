// pv_math.h -- fixed-size FP64 geometry for the gfx950 kernels and the host-side adapter code.
//
// Everything is a free inline function over plain double arrays (row-major 3x3, quaternion x,y,z,w) so
// that values live in VGPRs on the device; nothing here allocates or touches memory it was not handed.
// Semantics follow the reference's helpers (pvio/src/pvio/geometry/lie_algebra.h:25-42,
// lie_algebra.cpp:22-59) including the Eigen-3.3 AngleAxis conventions they inherit:
//   expmap(w)  = Quaternion(AngleAxis(|w|, w.stableNormalized()))     -> identity for w == 0
//   logmap(q)  = angle * axis with angle = 2 atan2(|v|, |w|), axis flipped when w < 0 (short way)
#pragma once
#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__) || defined(PV_HIPEMU)
#include <hip/hip_runtime.h>
#define PV_HD __host__ __device__ __forceinline__
#else
#define PV_HD inline
#endif

namespace pv {

PV_HD void v3_set(double *o, double a, double b, double c) { o[0] = a, o[1] = b, o[2] = c; }
PV_HD void v3_copy(double *o, const double *a) { o[0] = a[0], o[1] = a[1], o[2] = a[2]; }
PV_HD void v3_add(double *o, const double *a, const double *b) { o[0] = a[0] + b[0], o[1] = a[1] + b[1], o[2] = a[2] + b[2]; }
PV_HD void v3_sub(double *o, const double *a, const double *b) { o[0] = a[0] - b[0], o[1] = a[1] - b[1], o[2] = a[2] - b[2]; }
PV_HD void v3_axpy(double *o, double s, const double *x) { o[0] += s * x[0], o[1] += s * x[1], o[2] += s * x[2]; } // o += s x
PV_HD double v3_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
PV_HD void v3_cross(double *o, const double *a, const double *b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x, o[1] = y, o[2] = z;
}
PV_HD double v3_norm(const double *a) { return sqrt(v3_dot(a, a)); }

// 3x3 row-major
PV_HD void m3_mul(double *o, const double *a, const double *b) { // o = a b   (o must not alias)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
PV_HD void m3_mul_tn(double *o, const double *a, const double *b) { // o = a^T b
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
}
PV_HD void m3_vec(double *o, const double *a, const double *v) { // o = a v   (o must not alias v)
    o[0] = a[0] * v[0] + a[1] * v[1] + a[2] * v[2];
    o[1] = a[3] * v[0] + a[4] * v[1] + a[5] * v[2];
    o[2] = a[6] * v[0] + a[7] * v[1] + a[8] * v[2];
}
PV_HD void m3_tvec(double *o, const double *a, const double *v) { // o = a^T v
    o[0] = a[0] * v[0] + a[3] * v[1] + a[6] * v[2];
    o[1] = a[1] * v[0] + a[4] * v[1] + a[7] * v[2];
    o[2] = a[2] * v[0] + a[5] * v[1] + a[8] * v[2];
}
PV_HD void m3_transpose(double *o, const double *a) {
    o[0] = a[0], o[1] = a[3], o[2] = a[6], o[3] = a[1], o[4] = a[4], o[5] = a[7], o[6] = a[2], o[7] = a[5], o[8] = a[8];
}
PV_HD void m3_hat(double *o, const double *w) { // lie_algebra.h:25-30
    o[0] = 0, o[1] = -w[2], o[2] = w[1];
    o[3] = w[2], o[4] = 0, o[5] = -w[0];
    o[6] = -w[1], o[7] = w[0], o[8] = 0;
}
// o = a * hat(w): column j of the product is a x-ed by w per row: (a hat(w))_row = row x w ... expanded directly
PV_HD void m3_mul_hat(double *o, const double *a, const double *w) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a0 = a[3 * i], a1 = a[3 * i + 1], a2 = a[3 * i + 2];
        o[3 * i + 0] = a1 * w[2] - a2 * w[1];
        o[3 * i + 1] = a2 * w[0] - a0 * w[2];
        o[3 * i + 2] = a0 * w[1] - a1 * w[0];
    }
}
PV_HD void m3_inverse(double *o, const double *m) { // adjugate / determinant (Eigen's fixed 3x3 path)
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double id = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
    o[0] = c00 * id, o[3] = c01 * id, o[6] = c02 * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id, o[4] = (m[0] * m[8] - m[2] * m[6]) * id, o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id, o[5] = (m[2] * m[3] - m[0] * m[5]) * id, o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// quaternions, x y z w
// No FMA contraction here: q0^-1 (x) q0 must be EXACTLY (0,0,0,|q|^2) so that Log() of it is exactly zero -- the
// reference's first-time gauge prior multiplies this error by 1e15 (sliding_window_tracker.cpp:105-111), which
// would turn 1-ulp contraction noise into O(1e-3) of cost.
PV_HD void q_mul(double *o, const double *a, const double *b) { // o must not alias
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
PV_HD void q_conj(double *o, const double *a) { o[0] = -a[0], o[1] = -a[1], o[2] = -a[2], o[3] = a[3]; }
PV_HD void q_normalize(double *q) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n, q[1] /= n, q[2] /= n, q[3] /= n;
}
PV_HD void q_to_mat(double *R, const double *q) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
    R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
    R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}
PV_HD void q_rot(double *o, const double *q, const double *v) { // o = q v q^-1  (o must not alias v)
    double uv[3];
    v3_cross(uv, q, v);
    uv[0] += uv[0], uv[1] += uv[1], uv[2] += uv[2];
    double c[3];
    v3_cross(c, q, uv);
    o[0] = v[0] + q[3] * uv[0] + c[0], o[1] = v[1] + q[3] * uv[1] + c[1], o[2] = v[2] + q[3] * uv[2] + c[2];
}
PV_HD void q_rot_inv(double *o, const double *q, const double *v) {
    double qc[4];
    q_conj(qc, q);
    q_rot(o, qc, v);
}
PV_HD void q_expmap(double *q, const double *w) { // lie_algebra.h:32-37
    const double angle = v3_norm(w);
    const double mx = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
    double ax[3] = {w[0], w[1], w[2]};
    if (mx > 0.0) {
        const double s0 = w[0] / mx, s1 = w[1] / mx, s2 = w[2] / mx;
        const double z = s0 * s0 + s1 * s1 + s2 * s2;
        if (z > 0.0) {
            const double inv = 1.0 / sqrt(z);
            ax[0] = s0 * inv, ax[1] = s1 * inv, ax[2] = s2 * inv;
        }
    }
    double sh, ch;
    sincos(0.5 * angle, &sh, &ch); // one argument reduction for both (this sits on the critical path of every k_linearize prologue)
    q[0] = sh * ax[0], q[1] = sh * ax[1], q[2] = sh * ax[2], q[3] = ch;
}
PV_HD void q_logmap(double *w, const double *q) { // lie_algebra.h:39-42
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n != 0.0) {
        const double angle = 2.0 * atan2(n, fabs(q[3]));
        if (q[3] < 0) n = -n;
        w[0] = angle * (q[0] / n), w[1] = angle * (q[1] / n), w[2] = angle * (q[2] / n);
    } else {
        w[0] = w[1] = w[2] = 0.0;
    }
}
PV_HD void so3_right_jacobian(double *J, const double *w) { // lie_algebra.cpp:22-59
    const double root2_eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)
    const double root4_eps = 1.220703125e-04;         // sqrt(sqrt(DBL_EPSILON))
    const double qdrt720 = 5.180043974068832, qdrt5040 = 8.425726873265677;
    const double sqrt24 = 4.898979485566356, sqrt120 = 10.954451150103322;
    const double angle = v3_norm(w), angle2 = angle * angle;
    double cos_term, sin_term;
    if (angle > root4_eps * qdrt720) {
        cos_term = (1 - cos(angle)) / angle2;
    } else {
        cos_term = 0.5;
        if (angle > root2_eps * sqrt24) cos_term -= angle2 / 24.0;
    }
    if (angle > root4_eps * qdrt5040) {
        sin_term = (angle - sin(angle)) / (angle * angle2);
    } else {
        sin_term = 1.0 / 6.0;
        if (angle > root2_eps * sqrt120) sin_term -= angle2 / 120.0;
    }
    double H[9], H2[9];
    m3_hat(H, w);
    m3_mul(H2, H, H);
#pragma unroll
    for (int k = 0; k < 9; ++k) J[k] = -cos_term * H[k] + sin_term * H2[k];
    J[0] += 1.0, J[4] += 1.0, J[8] += 1.0;
}

// x (+) delta on one 16-double frame state (QuaternionParameterization::Plus, quaternion_parameterization.h:28-32)
PV_HD void pose_plus(double *out, const double *x, const double *dtheta, const double *dp) {
    double e[4], q[4];
    q_expmap(e, dtheta);
    q_mul(q, x, e);
    q_normalize(q);
    out[0] = q[0], out[1] = q[1], out[2] = q[2], out[3] = q[3];
    out[4] = x[4] + dp[0], out[5] = x[5] + dp[1], out[6] = x[6] + dp[2];
}

} // namespace pv
