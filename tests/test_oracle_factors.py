"""Finite-difference validation of the oracle's factor Jacobians.

The reference ships a (never instantiated) validator with the same idea:
pvio/src/pvio/estimation/ceres/cost_function_validator.h:183-323 (central differences through the local
parameterization).  Here: central differences, step 1e-6, through oracle_plus (q (+) theta).
"""
import ctypes as C

import numpy as np
import pytest

from pvio_amd import synth

dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(dp)


def rand_state(rng, scale=1.0):
    s = np.zeros(16)
    q = rng.normal(size=4)
    s[0:4] = q / np.linalg.norm(q)
    s[4:7] = rng.normal(size=3) * scale
    s[7:10] = rng.normal(size=3)
    s[10:13] = rng.normal(size=3) * 1e-2
    s[13:16] = rng.normal(size=3) * 1e-1
    return s


def plus(O, s, d15):
    out = np.zeros(16)
    O.lib().oracle_plus(_d(s), _d(np.ascontiguousarray(d15)), _d(out))
    return out


def fd_jacobian(f, x_states, O, h=1e-6):
    """f(list of states) -> residual vector.  Returns d r / d (15-dim tangent of each state), central diff."""
    r0 = f(x_states)
    J = np.zeros((r0.size, 15 * len(x_states)))
    for i in range(len(x_states)):
        for k in range(15):
            d = np.zeros(15)
            d[k] = h
            xp = list(x_states)
            xm = list(x_states)
            xp[i] = plus(O, x_states[i], d)
            xm[i] = plus(O, x_states[i], -d)
            J[:, 15 * i + k] = (f(xp) - f(xm)) / (2 * h)
    return J


def test_reprojection_jacobian(oracle):
    rng = np.random.default_rng(1)
    L = oracle.lib()
    for trial in range(20):
        st, sr = rand_state(rng, 0.3), rand_state(rng, 0.3)
        # keep the point in front of both cameras: small rotations around identity
        st[0:4] = synth.qexp(rng.normal(size=3) * 0.2)
        sr[0:4] = synth.qexp(rng.normal(size=3) * 0.2)
        cam_r = np.concatenate([synth.qexp(rng.normal(size=3) * 0.1), rng.normal(size=3) * 0.05])
        cam_t = np.concatenate([synth.qexp(rng.normal(size=3) * 0.1), rng.normal(size=3) * 0.05])
        z_ref, z_tgt = rng.normal(size=2) * 0.3, rng.normal(size=2) * 0.3
        W = np.array([400.0, 3.0, -2.0, 380.0])
        rho = 0.2 + rng.uniform() * 0.5

        def f(states, rho_=rho):
            r = np.zeros(2)
            L.oracle_eval_reprojection(_d(states[0]), _d(states[1]), float(rho_), _d(z_ref), _d(z_tgt), _d(cam_r), _d(cam_t), _d(W), _d(r), None)
            return r

        r = np.zeros(2)
        J = np.zeros((2, 13))
        L.oracle_eval_reprojection(_d(st), _d(sr), rho, _d(z_ref), _d(z_tgt), _d(cam_r), _d(cam_t), _d(W), _d(r), _d(J))
        Jfd = fd_jacobian(f, [st, sr], oracle)
        np.testing.assert_allclose(J[:, 0:6], Jfd[:, 0:6], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(J[:, 6:12], Jfd[:, 15:21], rtol=1e-6, atol=1e-5)
        assert np.abs(Jfd[:, 6:15]).max() == 0 and np.abs(Jfd[:, 21:30]).max() == 0
        h = 1e-7
        Jrho = (f([st, sr], rho + h) - f([st, sr], rho - h)) / (2 * h)
        np.testing.assert_allclose(J[:, 12], Jrho, rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(f([st, sr]), r)


def _random_preint(oracle, rng, bias):
    n = 40
    t = np.arange(n) * 0.005
    w = rng.normal(size=(n, 3)) * 0.3
    a = rng.normal(size=(n, 3)) * 1.0 + np.array([0, 0, 9.8])
    nz = dict(cov_w=np.eye(3) * synth.COV_G, cov_a=np.eye(3) * synth.COV_A, cov_bg=np.eye(3) * synth.COV_BG, cov_ba=np.eye(3) * synth.COV_BA)
    return (t, w, a, n * 0.005, nz), oracle.preintegrate(t, w, a, n * 0.005, bias[:3], bias[3:], nz)


@pytest.mark.parametrize("imu_offset", [False, True])
def test_preintegration_jacobian(oracle, imu_offset):
    """Every block matches finite differences, also with a non-zero IMU lever arm: the p_center_i in
    preintegration_error_cost.h:90 is exact (R_i^T(.. - p_i) = R_cs^T [R_ci^T(.. - p_ci) - p_cs]), so SURVEY
    App. D item 4 is not a quirk."""
    rng = np.random.default_rng(2)
    L = oracle.lib()
    for trial in range(5):
        si, sj = rand_state(rng), rand_state(rng)
        sj[0:4] = synth.qmul(si[0:4], synth.qexp(rng.normal(size=3) * 0.3))
        bias0 = si[10:16] + rng.normal(size=6) * np.array([1e-3] * 3 + [1e-2] * 3)
        _, (delta, cov, U, jac) = _random_preint(oracle, rng, bias0)
        imu_i = np.array([0, 0, 0, 1, 0, 0, 0], float)
        imu_j = imu_i.copy()
        if imu_offset:
            imu_i = np.concatenate([synth.qexp(rng.normal(size=3) * 0.1), rng.normal(size=3) * 0.05])
            imu_j = imu_i.copy()
        Uid = np.eye(15).ravel()  # identity weight so that blocks can be compared one by one

        def f(states):
            r = np.zeros(15)
            L.oracle_eval_preintegration(_d(states[0]), _d(states[1]), _d(bias0), _d(delta), _d(Uid), _d(jac), _d(imu_i), _d(imu_j), _d(r), None)
            return r

        r = np.zeros(15)
        J = np.zeros((15, 30))
        L.oracle_eval_preintegration(_d(si), _d(sj), _d(bias0), _d(delta), _d(Uid), _d(jac), _d(imu_i), _d(imu_j), _d(r), _d(J))
        Jfd = fd_jacobian(f, [si, sj], oracle)
        np.testing.assert_allclose(J, Jfd, rtol=2e-5, atol=2e-5)
        # weighting: J_U = U J, r_U = U r
        rU, JU = np.zeros(15), np.zeros((15, 30))
        L.oracle_eval_preintegration(_d(si), _d(sj), _d(bias0), _d(delta), _d(U), _d(jac), _d(imu_i), _d(imu_j), _d(rU), _d(JU))
        Um = U.reshape(15, 15)
        np.testing.assert_allclose(rU, Um @ r, rtol=1e-9, atol=1e-9 * np.abs(Um @ r).max())
        np.testing.assert_allclose(JU, Um @ J, rtol=1e-9, atol=1e-9 * np.abs(Um @ J).max())


def test_preintegrator_consistency(oracle):
    """sqrt_inv_cov^T sqrt_inv_cov == cov^-1, U upper triangular; bias Jacobians match finite differences
    of the re-integrated delta (preintegrator.cpp:69-75)."""
    rng = np.random.default_rng(3)
    bias = np.concatenate([rng.normal(size=3) * 1e-3, rng.normal(size=3) * 1e-2])
    (t, w, a, te, nz), (delta, cov, U, jac) = _random_preint(oracle, rng, bias)
    U = U.reshape(15, 15)
    cov = cov.reshape(15, 15)
    assert np.allclose(U, np.triu(U))
    np.testing.assert_allclose(U.T @ U @ cov, np.eye(15), atol=1e-6)
    J = jac.reshape(5, 3, 3)
    h = 1e-6
    for k in range(3):
        db = np.zeros(6)
        db[k] = h
        dpl = oracle.preintegrate(t, w, a, te, (bias + db)[:3], (bias + db)[3:], nz)[0]
        dmi = oracle.preintegrate(t, w, a, te, (bias - db)[:3], (bias - db)[3:], nz)[0]
        np.testing.assert_allclose((dpl[5:8] - dmi[5:8]) / (2 * h), J[1][:, k], rtol=1e-4, atol=1e-7)   # dp_dbg
        np.testing.assert_allclose((dpl[8:11] - dmi[8:11]) / (2 * h), J[3][:, k], rtol=1e-4, atol=1e-7)  # dv_dbg
        # dq_dbg: log(dq^-1 dq+) / h
        dq = synth.qmul(synth.qconj(delta[1:5]), dpl[1:5])
        np.testing.assert_allclose(2 * dq[:3] / h, J[0][:, k], rtol=1e-3, atol=1e-6)
        db = np.zeros(6)
        db[3 + k] = h
        dpl = oracle.preintegrate(t, w, a, te, bias[:3], (bias + db)[3:], nz)[0]
        dmi = oracle.preintegrate(t, w, a, te, bias[:3], (bias - db)[3:], nz)[0]
        np.testing.assert_allclose((dpl[5:8] - dmi[5:8]) / (2 * h), J[2][:, k], rtol=1e-4, atol=1e-7)   # dp_dba
        np.testing.assert_allclose((dpl[8:11] - dmi[8:11]) / (2 * h), J[4][:, k], rtol=1e-4, atol=1e-7)  # dv_dba


def test_prior_jacobian_and_identity(oracle):
    rng = np.random.default_rng(4)
    L = oracle.lib()
    n = 3
    lin = np.stack([rand_state(rng) for _ in range(n)])
    S = rng.normal(size=(15 * n, 15 * n))
    s = rng.normal(size=15 * n)
    # at the linearization point r == s (marginalization_error_cost.h:91)
    r = np.zeros(15 * n)
    L.oracle_eval_prior(n, _d(lin), _d(lin), _d(S), _d(s), _d(r), None)
    np.testing.assert_allclose(r, s, atol=1e-12)
    states = np.stack([plus(oracle, lin[i], rng.normal(size=15) * 0.05) for i in range(n)])
    J = np.zeros((15 * n, 15 * n))
    L.oracle_eval_prior(n, _d(states), _d(lin), _d(S), _d(s), _d(r), _d(J))

    def f(sts):
        rr = np.zeros(15 * n)
        L.oracle_eval_prior(n, _d(np.stack(sts)), _d(lin), _d(S), _d(s), _d(rr), None)
        return rr

    Jfd = fd_jacobian(f, [states[i] for i in range(n)], oracle)
    np.testing.assert_allclose(J, Jfd, rtol=1e-6, atol=1e-6)


def test_rotation_prior_jacobian_and_zero(oracle):
    """RotationPriorFactor (no reference counterpart): r = W Log(q0^-1 q), zero at q0, Jacobian by central differences, and the
    residual against a numpy restatement of the logarithm"""
    rng = np.random.default_rng(11)
    L = oracle.lib()
    for _ in range(5):
        x0 = rand_state(rng)
        W = rng.normal(size=(3, 3)) + 3 * np.eye(3)
        r = np.zeros(3)
        L.oracle_eval_rot_prior(_d(x0), _d(np.ascontiguousarray(x0[0:4])), _d(W), _d(r), None)
        np.testing.assert_allclose(r, 0, atol=1e-14)
        x = plus(oracle, x0, np.r_[rng.normal(size=3) * 0.3, np.zeros(12)])
        J = np.zeros((3, 3))
        L.oracle_eval_rot_prior(_d(x), _d(np.ascontiguousarray(x0[0:4])), _d(W), _d(r), _d(J))
        # numpy: dq = q0^-1 q, Log = 2 atan2(|v|, w) v / |v|
        a, b = x0[0:4] * np.array([-1, -1, -1, 1.0]), x[0:4]
        dq = np.r_[a[3] * b[:3] + b[3] * a[:3] + np.cross(a[:3], b[:3]), a[3] * b[3] - a[:3] @ b[:3]]
        if dq[3] < 0:
            dq = -dq
        nv = np.linalg.norm(dq[:3])
        np.testing.assert_allclose(r, W @ (2 * np.arctan2(nv, dq[3]) * dq[:3] / nv), rtol=1e-12, atol=1e-14)

        def f(sts):
            rr = np.zeros(3)
            L.oracle_eval_rot_prior(_d(sts[0]), _d(np.ascontiguousarray(x0[0:4])), _d(W), _d(rr), None)
            return rr

        Jfd = fd_jacobian(f, [x], oracle)
        np.testing.assert_allclose(J, Jfd[:, 0:3], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(Jfd[:, 3:], 0, atol=1e-9)


def test_plane_jacobian(oracle):
    rng = np.random.default_rng(5)
    L = oracle.lib()
    for K in (2, 4, 7):
        # a point on the plane n.x = d seen from K cameras
        n = np.array([0.1, -0.2, 1.0])
        n /= np.linalg.norm(n)
        d = 1.5
        X = np.array([0.3, -0.2, 0.0])
        X = X - (X @ n - d) * n
        states, cams, zs = [], [], []
        for k in range(K):
            s = rand_state(rng, 0.2)
            s[0:4] = synth.qexp(rng.normal(size=3) * 0.1)
            s[4:7] = rng.normal(size=3) * 0.3 + np.array([0, 0, -2.0])
            cam = np.concatenate([synth.qexp(rng.normal(size=3) * 0.05), rng.normal(size=3) * 0.03])
            Rwc = synth.qmat(synth.qmul(s[0:4], cam[0:4]))
            pwc = s[4:7] + synth.qmat(s[0:4]) @ cam[4:7]
            y = Rwc.T @ (X - pwc)
            zs.append(y[:2] / y[2] + rng.normal(size=2) * 1e-3)
            states.append(s)
            cams.append(cam)
        states, cams, zs = np.stack(states), np.stack(cams), np.stack(zs)
        r = np.zeros(1)
        J = np.zeros((K, 6))
        L.oracle_eval_plane(K, _d(states), _d(cams), _d(zs), _d(n), d, 100.0, _d(r), _d(J))

        def f(sts):
            rr = np.zeros(1)
            L.oracle_eval_plane(K, _d(np.stack(sts)), _d(cams), _d(zs), _d(n), d, 100.0, _d(rr), None)
            return rr

        Jfd = fd_jacobian(f, [states[k] for k in range(K)], oracle, h=1e-6)
        for k in range(K):
            np.testing.assert_allclose(J[k], Jfd[0, 15 * k:15 * k + 6], rtol=1e-4, atol=1e-4 * np.abs(J).max())


def test_lie_helpers(oracle):
    rng = np.random.default_rng(6)
    L = oracle.lib()
    for _ in range(50):
        w = rng.normal(size=3) * rng.choice([1e-9, 1e-4, 0.3, 2.5])
        if np.linalg.norm(w) > 3.0:
            w *= 3.0 / np.linalg.norm(w)
        q, w2 = np.zeros(4), np.zeros(3)
        L.oracle_expmap(_d(w), _d(q))
        assert abs(np.linalg.norm(q) - 1) < 1e-14
        L.oracle_logmap(_d(q), _d(w2))
        np.testing.assert_allclose(w2, w, rtol=1e-9, atol=1e-15)
        # right Jacobian: exp(w + d) ~= exp(w) exp(Jr d)
        Jr = np.zeros(9)
        L.oracle_right_jacobian(_d(w), _d(Jr))
        Jr = Jr.reshape(3, 3)
        d = rng.normal(size=3) * 1e-6
        q2 = np.zeros(4)
        L.oracle_expmap(_d(w + d), _d(q2))
        dq = synth.qmul(synth.qconj(q), q2)
        np.testing.assert_allclose(2 * dq[:3], Jr @ d, rtol=1e-4, atol=1e-11)
    # identity and zero
    q = np.zeros(4)
    L.oracle_expmap(_d(np.zeros(3)), _d(q))
    assert (q == np.array([0, 0, 0, 1.0])).all()
    w = np.ones(3)
    L.oracle_logmap(_d(np.array([0, 0, 0, 1.0])), _d(w))
    assert (w == 0).all()
    # short-way log for w < 0
    L.oracle_logmap(_d(np.array([0.1, 0, 0, -np.sqrt(1 - 0.01)])), _d(w))
    assert w[0] < 0 and abs(w[0]) < 0.3
