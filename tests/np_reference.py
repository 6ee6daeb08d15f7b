"""Independent dense numpy implementation of the Ceres-1.14 Dogleg trust-region loop.

Second opinion for the C++ oracle's solver algebra: it shares only the single-factor evaluators with the
oracle (oracle_eval_*); the Jacobian is assembled densely, the Gauss-Newton step is a plain dense solve
of (J^T J + mu D^2) y = J^T r -- no Schur complement, no block structure -- and the bookkeeping
(Jacobi scaling, Cauchy point, dogleg interpolation, tolerances, live-bias update) is written
independently.  Small problems only.
"""
import ctypes as C

import numpy as np

dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(dp)


class DenseProblem:
    def __init__(self, pb, O):
        self.pb, self.O, self.L = pb, O, O.lib()
        lin = O.linearize(pb, pb.frame_state, pb.lm_inv_depth)
        self.pose_off, self.motion_off, self.P = lin["pose_off"], lin["motion_off"], lin["P"]
        self.lm_used = np.diff(pb.lm_obs_ptr) > 0
        self.lm_col = np.full(pb.n_landmarks, -1)
        self.lm_col[self.lm_used] = self.P + np.arange(self.lm_used.sum())
        self.ncols = self.P + int(self.lm_used.sum())

    def frame_cols(self, f):
        cols = np.full(15, -1)
        if self.pose_off[f] >= 0:
            cols[0:6] = self.pose_off[f] + np.arange(6)
        if self.motion_off[f] >= 0:
            cols[6:15] = self.motion_off[f] + np.arange(9)
        return cols

    def evaluate(self, fs, rho, user, jac=True):
        pb, L = self.pb, self.L
        rows_r, rows_J = [], []
        cost = 0.0

        def push(r, Jblocks, robust):
            nonlocal cost
            s = float(r @ r)
            if robust:
                cost += 0.5 * np.log1p(s)
                sw = np.sqrt(1.0 / (1.0 + s))
            else:
                cost += 0.5 * s
                sw = 1.0
            if jac:
                Jrow = np.zeros((r.size, self.ncols))
                for cols, Jb in Jblocks:
                    m = cols >= 0
                    Jrow[:, cols[m]] += Jb[:, m]
                rows_J.append(sw * Jrow)
            rows_r.append(sw * r)

        if pb.prior_frames.shape[0] > 0:
            n = pb.prior_frames.shape[0]
            st = np.ascontiguousarray(fs[pb.prior_frames])
            r = np.zeros(15 * n)
            J = np.zeros((15 * n, 15 * n))
            L.oracle_eval_prior(n, _d(st), _d(pb.prior_lin_state), _d(pb.prior_S), _d(pb.prior_s), _d(r), _d(J) if jac else None)
            push(r, [(self.frame_cols(f), J[:, 15 * i:15 * i + 15]) for i, f in enumerate(pb.prior_frames)], False)
        for i, f in enumerate(getattr(pb, "rot_prior_frame", [])):  # RotationPriorFactor (no reference counterpart), no loss
            if self.pose_off[f] < 0:
                continue
            r, J = np.zeros(3), np.zeros((3, 3))
            L.oracle_eval_rot_prior(_d(np.ascontiguousarray(fs[f])), _d(pb.rot_prior_q0[i]), _d(pb.rot_prior_sqrt_info[i]), _d(r), _d(J) if jac else None)
            push(r, [(self.frame_cols(f)[:3], J)], False)
        for l in range(pb.n_landmarks):
            a = pb.lm_anchor_frame[l]
            for o in range(pb.lm_obs_ptr[l], pb.lm_obs_ptr[l + 1]):
                t = pb.obs_frame[o]
                r = np.zeros(2)
                J = np.zeros((2, 13))
                L.oracle_eval_reprojection(_d(fs[t]), _d(fs[a]), float(rho[l]), _d(pb.lm_anchor_z[l]), _d(pb.obs_z[o]),
                                           _d(pb.cam_extrinsic[a]), _d(pb.cam_extrinsic[t]), _d(pb.sqrt_inv_cov[t]), _d(r), _d(J) if jac else None)
                ct, ca = self.frame_cols(t)[:6], self.frame_cols(a)[:6]
                push(r, [(ct, J[:, 0:6]), (ca, J[:, 6:12]), (np.array([self.lm_col[l]]), J[:, 12:13])], True)
        for f in range(pb.n_plane_factors):
            b, e = pb.plane_obs_ptr[f], pb.plane_obs_ptr[f + 1]
            frames = pb.plane_obs_frame[b:e]
            if all(pb.frame_fixed[frames]):
                continue
            K = e - b
            st = np.ascontiguousarray(fs[frames])
            cams = np.ascontiguousarray(pb.cam_extrinsic[frames])
            z = np.ascontiguousarray(pb.plane_obs_z[b:e])
            r = np.zeros(1)
            J = np.zeros((K, 6))
            L.oracle_eval_plane(int(K), _d(st), _d(cams), _d(z), _d(pb.plane_normal[f]), float(pb.plane_distance[f]),
                                float(pb.plane_sqrt_inv_cov), _d(r), _d(J) if jac else None)
            push(r, [(self.frame_cols(fr)[:6], J[k:k + 1, :]) for k, fr in enumerate(frames)], True)
        if pb.use_inertial:
            for j in range(1, pb.n_frames):
                if not pb.preint_valid[j]:
                    continue
                i = j - 1
                r = np.zeros(15)
                J = np.zeros((15, 30))
                bias0 = np.ascontiguousarray(user[i, 10:16])
                L.oracle_eval_preintegration(_d(fs[i]), _d(fs[j]), _d(bias0), _d(pb.preint_delta[j]), _d(pb.preint_sqrt_inv_cov[j]),
                                             _d(pb.preint_jacobian[j]), _d(pb.imu_extrinsic[i]), _d(pb.imu_extrinsic[j]), _d(r), _d(J) if jac else None)
                push(r, [(self.frame_cols(i), J[:, 0:15]), (self.frame_cols(j), J[:, 15:30])], False)
        r = np.concatenate(rows_r)
        J = np.concatenate(rows_J) if jac else None
        return cost, r, J

    def plus(self, fs, rho, delta):
        fs2, rho2 = fs.copy(), rho.copy()
        for f in range(self.pb.n_frames):
            cols = self.frame_cols(f)
            d = np.where(cols >= 0, delta[np.maximum(cols, 0)], 0.0)
            out = np.zeros(16)
            self.L.oracle_plus(_d(np.ascontiguousarray(fs[f])), _d(np.ascontiguousarray(d)), _d(out))
            # only free blocks move (oracle_plus renormalizes q even for a zero step: keep fixed blocks bit-identical)
            if cols[0] >= 0:
                fs2[f, 0:7] = out[0:7]
            if cols[6] >= 0:
                fs2[f, 7:16] = out[7:16]
        m = self.lm_used
        rho2[m] = rho[m] + delta[self.lm_col[m]]
        return fs2, rho2

    def ambient(self, fs, rho):
        parts = []
        for f in range(self.pb.n_frames):
            if self.pose_off[f] >= 0:
                parts.append(fs[f, 0:7])
            if self.motion_off[f] >= 0:
                parts.append(fs[f, 7:16])
        parts.append(rho[self.lm_used])
        return np.concatenate(parts)


def solve(pb, O, state_update=True):
    """Returns (trace list of dicts, final frame_state, final rho, termination)."""
    return solve_dense(DenseProblem(pb, O), pb.frame_state.copy(), pb.lm_inv_depth.copy(), pb.max_iterations, state_update)


def solve_dense(D, fs, rho, max_iterations, state_update=True):
    """The loop itself, for any problem object with evaluate(fs, rho, user, jac) / plus / ambient / ncols."""
    user = fs.copy()
    best = (fs.copy(), rho.copy())
    trace = []
    radius, mu, reuse, invalid = 1e4, 1e-8, False, 0
    x_cost, r, J = D.evaluate(fs, rho, user)
    g_unscaled = J.T @ r
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    J = J * scale

    def gmax(fs_, rho_, g_):
        f2, r2 = D.plus(fs_, rho_, -g_)
        return np.abs(D.ambient(fs_, rho_) - D.ambient(f2, r2)).max() if D.ncols else 0.0

    grad_max = gmax(fs, rho, g_unscaled)
    x_norm = np.linalg.norm(D.ambient(fs, rho))
    min_cost = np.inf
    it, success, term = 0, True, 1
    rec = dict(cost=x_cost, cost_change=0.0, step_norm=0.0, relative_decrease=0.0, step_is_valid=1)
    diag = ghat = gn = None
    alpha = step_norm_dl = 0.0
    while True:
        if success and x_cost < min_cost:
            min_cost = x_cost
            best = (fs.copy(), rho.copy())
        trace.append(dict(rec, iteration=it, step_is_successful=int(success), gradient_max_norm=grad_max, trust_region_radius=radius, mu=mu,
                          state=np.concatenate([fs.ravel(), rho])))
        if success and state_update:
            user = best[0].copy()
        if it >= max_iterations:
            term = 1
            break
        if success and grad_max <= 1e-10:
            term = 0
            break
        if radius <= 1e-32:
            term = 0
            break
        it += 1
        success = False
        rec = dict(cost=x_cost, cost_change=0.0, step_norm=0.0, relative_decrease=0.0, step_is_valid=0)
        ok = True
        if not reuse:
            reuse = True
            diag = np.sqrt(np.clip((J * J).sum(0), 1e-6, 1e32))
            ghat = (J.T @ r) / diag
            Jg = J @ (ghat / diag)
            alpha = (ghat @ ghat) / (Jg @ Jg)
            ok = False
            while mu < 1.0:
                A = J.T @ J + np.diag(mu * diag * diag)
                try:
                    Lc = np.linalg.cholesky(A)
                    y = np.linalg.solve(Lc.T, np.linalg.solve(Lc, J.T @ r))
                    if np.all(np.isfinite(y)):
                        ok = True
                        break
                except np.linalg.LinAlgError:
                    pass
                mu *= 10.0
            if ok:
                gn = -diag * y
        if ok:
            gnorm, gnn = np.linalg.norm(ghat), np.linalg.norm(gn)
            if gnn <= radius:
                step, step_norm_dl = gn.copy(), gnn
            elif gnorm * alpha >= radius:
                step, step_norm_dl = -(radius / gnorm) * ghat, radius
            else:
                b_dot_a = -alpha * (ghat @ gn)
                a2 = (alpha * gnorm) ** 2
                bma2 = a2 - 2 * b_dot_a + gnn ** 2
                c = b_dot_a - a2
                d = np.sqrt(c * c + bma2 * (radius ** 2 - a2))
                beta = (d - c) / bma2 if c <= 0 else (radius ** 2 - a2) / (d + c)
                step = (-alpha * (1 - beta)) * ghat + beta * gn
                step_norm_dl = np.linalg.norm(step)
            step = step / diag
            mr = J @ step
            model_change = -mr @ (r + mr / 2.0)
            rec["step_is_valid"] = int(model_change > 0)
        if not ok or model_change <= 0:
            invalid += 1
            if invalid >= 5:
                term = 2
                break
            mu *= 10.0
            reuse = False
            continue
        invalid = 0
        delta = step * scale
        cfs, crho = D.plus(fs, rho, delta)
        cand_cost, _, _ = D.evaluate(cfs, crho, user, jac=False)
        if not np.isfinite(cand_cost):
            cand_cost = np.finfo(float).max
        rec["step_norm"] = np.linalg.norm(D.ambient(fs, rho) - D.ambient(cfs, crho))
        if rec["step_norm"] <= 1e-8 * (x_norm + 1e-8):
            term = 0
            break
        rec["cost_change"] = x_cost - cand_cost
        if abs(rec["cost_change"]) <= 1e-6 * x_cost:
            term = 0
            break
        rel = rec["cost_change"] / model_change
        rec["relative_decrease"] = rel
        if rel > 1e-3:
            fs, rho = cfs, crho
            x_norm = np.linalg.norm(D.ambient(fs, rho))
            x_cost, r, J = D.evaluate(fs, rho, user)
            g_unscaled = J.T @ r
            J = J * scale
            grad_max = gmax(fs, rho, g_unscaled)
            success = True
            rec["cost"] = x_cost
            if rel < 0.25:
                radius *= 0.5
            if rel > 0.75:
                radius = max(radius, 3.0 * step_norm_dl)
            mu = max(1e-8, 2.0 * mu / 10.0)
            reuse = False
        else:
            radius *= 0.5
            reuse = True
            rec["cost"] = cand_cost
    return trace, best[0], best[1], term, it


def marginalize(pb, O, fs, rho, victim):
    """Information matrix / vector of the prior left by BundleAdjustor::marginalize_frame (bundle_adjustor.cpp:348-599),
    restated from the dense Jacobian: stack every factor that touches the victim (old prior, the one or two
    pre-integration factors, every reprojection factor of a landmark the victim sees; no robust loss, every frame a
    free 15-dof column block), form J^T J / J^T r and take the Schur complement over {those landmarks, the victim}.
    Independent of the oracle's hand-indexed block accumulation; only the single-factor evaluators are shared."""
    L = O.lib()
    N = pb.n_frames
    seen = []
    for l in range(pb.n_landmarks):
        fr = [pb.lm_anchor_frame[l]] + list(pb.obs_frame[pb.lm_obs_ptr[l]:pb.lm_obs_ptr[l + 1]])
        if victim in fr and pb.lm_obs_ptr[l + 1] > pb.lm_obs_ptr[l]:
            seen.append(l)
    lm_col = {l: 15 * N + k for k, l in enumerate(seen)}
    ncols = 15 * N + len(seen)
    rows_r, rows_J = [], []
    if pb.prior_frames.shape[0] > 0:
        n = pb.prior_frames.shape[0]
        st = np.ascontiguousarray(fs[pb.prior_frames])
        r, J = np.zeros(15 * n), np.zeros((15 * n, 15 * n))
        L.oracle_eval_prior(n, _d(st), _d(pb.prior_lin_state), _d(pb.prior_S), _d(pb.prior_s), _d(r), _d(J))
        Jr = np.zeros((15 * n, ncols))
        for i, f in enumerate(pb.prior_frames):
            Jr[:, 15 * f:15 * f + 15] += J[:, 15 * i:15 * i + 15]
        rows_r.append(r), rows_J.append(Jr)
    for i, f in enumerate(getattr(pb, "rot_prior_frame", [])):  # the victim's rotation prior goes into the new prior
        if f != victim:
            continue
        r, J = np.zeros(3), np.zeros((3, 3))
        L.oracle_eval_rot_prior(_d(np.ascontiguousarray(fs[f])), _d(pb.rot_prior_q0[i]), _d(pb.rot_prior_sqrt_info[i]), _d(r), _d(J))
        Jr = np.zeros((3, ncols))
        Jr[:, 15 * f:15 * f + 3] = J
        rows_r.append(r), rows_J.append(Jr)
    if pb.use_inertial:
        for j in (victim, victim + 1):
            if j <= 0 or j >= N or not pb.preint_valid[j]:
                continue
            i = j - 1
            r, J = np.zeros(15), np.zeros((15, 30))
            bias0 = np.ascontiguousarray(fs[i, 10:16])  # the live bias is the linearization bias here (:416-450)
            L.oracle_eval_preintegration(_d(fs[i]), _d(fs[j]), _d(bias0), _d(pb.preint_delta[j]), _d(pb.preint_sqrt_inv_cov[j]),
                                         _d(pb.preint_jacobian[j]), _d(pb.imu_extrinsic[i]), _d(pb.imu_extrinsic[j]), _d(r), _d(J))
            Jr = np.zeros((15, ncols))
            Jr[:, 15 * i:15 * i + 30] = J
            rows_r.append(r), rows_J.append(Jr)
    for l in seen:
        a = pb.lm_anchor_frame[l]
        for o in range(pb.lm_obs_ptr[l], pb.lm_obs_ptr[l + 1]):
            t = pb.obs_frame[o]
            r, J = np.zeros(2), np.zeros((2, 13))
            L.oracle_eval_reprojection(_d(fs[t]), _d(fs[a]), float(rho[l]), _d(pb.lm_anchor_z[l]), _d(pb.obs_z[o]),
                                       _d(pb.cam_extrinsic[a]), _d(pb.cam_extrinsic[t]), _d(pb.sqrt_inv_cov[t]), _d(r), _d(J))
            Jr = np.zeros((2, ncols))
            Jr[:, 15 * t:15 * t + 6] += J[:, 0:6]
            Jr[:, 15 * a:15 * a + 6] += J[:, 6:12]
            Jr[:, lm_col[l]] += J[:, 12]
            rows_r.append(r), rows_J.append(Jr)
    r, J = np.concatenate(rows_r), np.concatenate(rows_J)
    H, b = J.T @ J, J.T @ r
    F = 15 * N
    # landmarks first (diagonal block), then the victim's 15x15 block: the same elimination order as the reference,
    # which matters numerically when the first-time gauge prior (information ~1e30) sits on the victim
    Hll = np.diag(H)[F:]
    ok = np.isfinite(1.0 / Hll)
    W = H[:F, F:][:, ok]
    Hf = H[:F, :F] - (W / Hll[ok]) @ W.T
    bf = b[:F] - (W / Hll[ok]) @ b[F:][ok]
    v = np.arange(15 * victim, 15 * victim + 15)
    rest = np.setdiff1d(np.arange(F), v)
    Hvv_inv = np.linalg.inv(Hf[np.ix_(v, v)])
    T = Hf[np.ix_(rest, v)] @ Hvv_inv
    return Hf[np.ix_(rest, rest)] - T @ Hf[np.ix_(v, rest)], bf[rest] - T @ bf[v]
