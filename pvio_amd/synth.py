"""Deterministic synthetic sliding windows for parity tests and bench.py.

Camera / IMU model: reference `config/euroc.yaml` (intrinsics :15, q_bc/p_bc :17-18, IMU at the body
origin :44-45, keypoint noise 0.5 px^2 :11-14, IMU noise densities :23-42).  Seed 648 = Config::random()
(pvio/src/pvio/config.cpp:91-93).  The PRNG is a counter-based SplitMix64 (+ Box-Muller) so the same
stream can be produced from any language without depending on numpy's generator internals.
"""
import numpy as np

from .problem import BAProblem

SEED = 648
K_EUROC = np.array([458.654, 457.296, 367.215, 248.375])
Q_BC = np.array([-7.7071797555374275e-03, 1.0499323370587278e-02, 7.0175280029197162e-01, 7.1230146066895372e-01])
P_BC = np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])
COV_G, COV_A, COV_BG, COV_BA = 2.8791302399999997e-08, 4.0e-6, 3.7608844899999997e-10, 9.0e-6
KEYPOINT_COV = 0.5
IMG_W, IMG_H, BORDER = 752, 480, 20
GRAVITY = 9.80665


class Rng:
    """Counter-based SplitMix64: value(i) = mix(seed + (i+1) * golden)."""

    def __init__(self, seed):
        self.seed = np.uint64(seed)
        self.ctr = 0

    def u64(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(self.ctr + 1, self.ctr + n + 1, dtype=np.uint64)
            z = self.seed + idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        self.ctr += n
        return z

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, n):
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        out = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return out[:n]


# ---- quaternion helpers (x, y, z, w) ----------------------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], -1)


def qconj(q):
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def qmat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def qexp(w):
    a = np.linalg.norm(w)
    if a < 1e-300:
        return np.array([0.0, 0.0, 0.0, 1.0])
    return np.concatenate([np.sin(a / 2) * w / a, [np.cos(a / 2)]])


def mat2q(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
    x = np.copysign(x, R[2, 1] - R[1, 2])
    y = np.copysign(y, R[0, 2] - R[2, 0])
    z = np.copysign(z, R[1, 0] - R[0, 1])
    q = np.array([x, y, z, w])
    return q / np.linalg.norm(q)


# ---- trajectory: orbit of radius 3 m looking at the centre, 0.3 rad/s, +-0.2 m vertical sinusoid ------
RADIUS, OMEGA, H_AMP, H_FREQ = 3.0, 0.3, 0.2, 0.8


def _pose(t):
    th = OMEGA * t
    p = np.array([RADIUS * np.cos(th), RADIUS * np.sin(th), H_AMP * np.sin(H_FREQ * t)])
    v = np.array([-RADIUS * OMEGA * np.sin(th), RADIUS * OMEGA * np.cos(th), H_AMP * H_FREQ * np.cos(H_FREQ * t)])
    acc = np.array([-RADIUS * OMEGA ** 2 * np.cos(th), -RADIUS * OMEGA ** 2 * np.sin(th), -H_AMP * H_FREQ ** 2 * np.sin(H_FREQ * t)])
    # EuRoC body axes: x up, y right, z forward (optical axis ~ body z); forward points at the orbit centre
    up = np.array([0.0, 0.0, 1.0])
    fwd = np.array([-np.cos(th), -np.sin(th), 0.0])
    right = np.cross(fwd, up)
    R = np.stack([up, right, fwd], 1)
    return R, p, v, acc


def make_window(n_frames=10, n_landmarks=1000, use_inertial=False, visibility=None, plane_fraction=0.0,
                seed=SEED, preintegrate=None, kf_dt=0.25, imu_rate=200.0, perturb=True, max_iterations=10,
                bias_init="near_truth", perturb_scale=None, plane_outliers=0, plane_outlier_offset=0.3, rot_prior_frames=(), duplicate_fraction=0.0):
    """Builds a BAProblem.  `preintegrate(t, w, a, t_end, bg, ba, noise_dict) -> (delta11, cov225, U225, jac45)`
    is required when use_inertial (the product's pvio_preintegrate or the oracle's).

    bias_init: "near_truth" (default; truth + N(0, 1e-5 rad/s / 1e-4 m/s^2), what a steady-state window holds)
    or "zero" (SURVEY 8d's original choice).  With "zero" the reference's live-bias read
    (preintegration_error_cost.h:57-58 + update_state_every_iteration) makes every step after the first accepted
    one see a different cost function and get rejected -- kept as a parity case, not as the benchmark."""
    N, M = n_frames, n_landmarks
    rng = Rng(seed)
    pb = BAProblem(N)
    pb.use_inertial = bool(use_inertial)
    pb.max_iterations = max_iterations
    pb.cam_extrinsic[:] = np.concatenate([Q_BC / np.linalg.norm(Q_BC), P_BC])
    pb.imu_extrinsic[:] = np.array([0, 0, 0, 1, 0, 0, 0], float)
    sw = np.array([K_EUROC[0] / np.sqrt(KEYPOINT_COV), 0.0, 0.0, K_EUROC[1] / np.sqrt(KEYPOINT_COV)])
    pb.sqrt_inv_cov[:] = sw
    pb.intrinsics[:] = K_EUROC
    q_bc = Q_BC / np.linalg.norm(Q_BC)
    R_bc = qmat(q_bc)

    # ground truth states
    truth = np.zeros((N, 16))
    bg_true, ba_true = np.full(3, 2e-3), np.full(3, 2e-2)
    Rs, ps = [], []
    for i in range(N):
        R, p, v, _ = _pose(i * kf_dt)
        truth[i, 0:4] = mat2q(R)
        truth[i, 4:7] = p
        truth[i, 7:10] = v
        truth[i, 10:13] = bg_true
        truth[i, 13:16] = ba_true
        Rs.append(R)
        ps.append(p)
    R_wc = [Rs[i] @ R_bc for i in range(N)]
    p_wc = [ps[i] + Rs[i] @ P_BC for i in range(N)]

    # landmarks: rejection-sample points around the orbit centre that project inside every frame
    pts = np.zeros((0, 3))
    while pts.shape[0] < M:
        n_try = max(256, 2 * (M - pts.shape[0]))
        u = rng.uniform(3 * n_try).reshape(n_try, 3)
        cand = np.stack([(u[:, 0] - 0.5) * 5.0, (u[:, 1] - 0.5) * 5.0, (u[:, 2] - 0.5) * 2.4], 1)
        ok = np.ones(n_try, bool)
        for i in range(N):
            y = (cand - p_wc[i]) @ R_wc[i]
            z = y[:, 2]
            with np.errstate(divide="ignore", invalid="ignore"):
                px = K_EUROC[0] * y[:, 0] / z + K_EUROC[2]
                py = K_EUROC[1] * y[:, 1] / z + K_EUROC[3]
            ok &= (z > 1.0) & (z < 8.0) & (px >= BORDER + 2) & (px < IMG_W - BORDER - 2) & (py >= BORDER + 2) & (py < IMG_H - BORDER - 2)
        pts = np.concatenate([pts, cand[ok]])
    pts = pts[:M]

    # plane variant: move a fraction of the landmarks onto two planes (A6 path)
    n_plane = int(round(plane_fraction * M))
    plane_defs = []
    if n_plane > 0:
        plane_defs = [(np.array([0.0, 0.0, 1.0]), -0.6), (np.array([1.0, 0.0, 0.0]) , 0.3)]
        half = n_plane // 2
        for k, (nrm, dist) in enumerate(plane_defs):
            sl = slice(0, half) if k == 0 else slice(half, n_plane)
            pts[sl] = pts[sl] - np.outer(pts[sl] @ nrm - dist, nrm)
        # a few plane tracks whose true point sits OFF its plane (what the post-solve re-validation of
        # bundle_adjustor.cpp:251-275 exists for): the first `plane_outliers` tracks of each plane
        for k, (nrm, dist) in enumerate(plane_defs):
            o = 0 if k == 0 else half
            pts[o:o + plane_outliers] += plane_outlier_offset * nrm

    # visibility: landmark l is seen in frames [s_l, s_l + K) (contiguous run); anchor = first
    if visibility is None or visibility >= N:
        start = np.zeros(M, int)
        K = N
    else:
        K = int(visibility)
        start = (rng.uniform(M) * (N - K + 1)).astype(int)
    noise = rng.normal(2 * M * N).reshape(M, N, 2) * np.sqrt(KEYPOINT_COV)
    z_all = np.zeros((M, N, 2))
    for i in range(N):
        y = (pts - p_wc[i]) @ R_wc[i]
        px = K_EUROC[0] * y[:, 0] / y[:, 2] + K_EUROC[2] + noise[:, i, 0]
        py = K_EUROC[1] * y[:, 1] / y[:, 2] + K_EUROC[3] + noise[:, i, 1]
        z_all[:, i, 0] = (px - K_EUROC[2]) / K_EUROC[0]
        z_all[:, i, 1] = (py - K_EUROC[3]) / K_EUROC[1]
    depth_anchor = np.array([((pts[l] - p_wc[start[l]]) @ R_wc[start[l]])[2] for l in range(M)])

    is_plane = np.zeros(M, bool)
    is_plane[:n_plane] = True
    lm_idx = np.nonzero(~is_plane)[0]
    Ml = lm_idx.shape[0]
    pb.lm_anchor_frame = start[lm_idx].astype(np.int32)
    pb.lm_anchor_z = z_all[lm_idx, start[lm_idx]]
    pb.lm_obs_ptr = (np.arange(Ml + 1) * (K - 1)).astype(np.int32)
    obs_frame = (start[lm_idx][:, None] + 1 + np.arange(K - 1)[None, :])
    pb.obs_frame = obs_frame.ravel().astype(np.int32)
    pb.obs_z = z_all[lm_idx[:, None], obs_frame].reshape(-1, 2)
    truth_rho = 1.0 / depth_anchor
    pb.truth_inv_depth = truth_rho[lm_idx].copy()

    if n_plane > 0:
        pidx = np.nonzero(is_plane)[0]
        pb.plane_obs_ptr = (np.arange(n_plane + 1) * K).astype(np.int32)
        pf = (start[pidx][:, None] + np.arange(K)[None, :])
        pb.plane_obs_frame = pf.ravel().astype(np.int32)
        pb.plane_obs_z = z_all[pidx[:, None], pf].reshape(-1, 2)
        half = n_plane // 2
        pb.plane_normal = np.array([plane_defs[0][0] if k < half else plane_defs[1][0] for k in range(n_plane)])
        pb.plane_distance = np.array([plane_defs[0][1] if k < half else plane_defs[1][1] for k in range(n_plane)])
        pb.plane_sqrt_inv_cov = np.sqrt(1.0 / 1.0e-4)  # pvio-pc/config/euroc.yaml plane.noise

    # IMU
    bias_guess = np.zeros((N, 6))
    if use_inertial and bias_init == "near_truth":
        bias_guess[:, :3] = bg_true + rng.normal(3 * N).reshape(N, 3) * 1e-5
        bias_guess[:, 3:] = ba_true + rng.normal(3 * N).reshape(N, 3) * 1e-4
    if use_inertial:
        assert preintegrate is not None, "use_inertial needs a preintegrate callable"
        dt_imu = 1.0 / imu_rate
        n_s = int(round(kf_dt * imu_rate))
        noise_d = dict(cov_w=np.eye(3) * COV_G, cov_a=np.eye(3) * COV_A, cov_bg=np.eye(3) * COV_BG, cov_ba=np.eye(3) * COV_BA)
        pb.meta["imu"] = []
        for j in range(1, N):
            t0 = (j - 1) * kf_dt
            ts = t0 + np.arange(n_s) * dt_imu
            w = np.zeros((n_s, 3))
            a = np.zeros((n_s, 3))
            nw = rng.normal(3 * n_s).reshape(n_s, 3) * np.sqrt(COV_G * imu_rate)
            na = rng.normal(3 * n_s).reshape(n_s, 3) * np.sqrt(COV_A * imu_rate)
            for k in range(n_s):
                R, _, _, acc = _pose(ts[k] + 0.5 * dt_imu)
                w[k] = np.array([OMEGA, 0.0, 0.0]) + bg_true + nw[k]  # body x = world z
                a[k] = R.T @ (acc + np.array([0, 0, GRAVITY])) + ba_true + na[k]
            # integrated at the INITIAL-GUESS biases of frame j-1, as solve() does at :224
            delta, cov, U, jac = preintegrate(ts, w, a, j * kf_dt, bias_guess[j - 1, :3], bias_guess[j - 1, 3:], noise_d)
            pb.preint_valid[j] = 1
            pb.preint_delta[j] = delta
            pb.preint_sqrt_inv_cov[j] = U
            pb.preint_jacobian[j] = jac
            pb.meta["imu"].append((ts, w, a, j * kf_dt))
        pb.meta["imu_noise"] = noise_d

    # initial guess
    init = truth.copy()
    rho0 = pb.truth_inv_depth.copy()
    if perturb:
        # vision-only: 0.5 deg / 2 cm / 5 %.  VIO: one tenth of that by default -- a steady-state
        # window starts from the previous optimum + an IMU-predicted new frame, and the pre-integration factors
        # are ~100x tighter than 2 cm over 0.25 s.
        ps_ = perturb_scale if perturb_scale is not None else (0.1 if use_inertial else 1.0)
        rot = rng.normal(3 * N).reshape(N, 3) * np.deg2rad(0.5) * ps_
        pos = rng.normal(3 * N).reshape(N, 3) * 0.02 * ps_
        vel = rng.normal(3 * N).reshape(N, 3) * 0.02 * ps_
        first = 0 if use_inertial else 1
        for i in range(first, N):
            init[i, 0:4] = qmul(truth[i, 0:4], qexp(rot[i]))
            init[i, 0:4] /= np.linalg.norm(init[i, 0:4])
            init[i, 4:7] += pos[i]
            init[i, 7:10] += vel[i]
        rho0 = rho0 * (1.0 + 0.05 * ps_ * rng.normal(Ml))
    init[:, 10:16] = bias_guess
    if not use_inertial:
        init[:, 7:10] = 0.0
        pb.frame_fixed[0] = 1  # initializer.cpp:199: frame 0 FF_FIX_POSE for the visual BA
    else:
        # first-time gauge prior: 1e15 on frame-0 q,p passed as sqrt-information, over frames 0..N-2
        # (sliding_window_tracker.cpp:100-112)
        n = N - 1
        S = np.zeros((15 * n, 15 * n))
        S[0:3, 0:3] = 1.0e15 * np.eye(3)
        S[3:6, 3:6] = 1.0e15 * np.eye(3)
        pb.prior_frames = np.arange(n, dtype=np.int32)
        pb.prior_S = S
        pb.prior_s = np.zeros(15 * n)
        pb.prior_lin_state = init[:n].copy()
    if len(rot_prior_frames) > 0:
        # RotationPriorFactor (no reference counterpart): "an attitude measurement" of the listed frames -- truth turned by
        # 0.3 degrees of noise, sqrt-information of ~0.3 degrees with a little cross-coupling.  Own random stream: the rest
        # of the window is the same with and without them.
        rr = Rng(seed + 4099)
        nrp = len(rot_prior_frames)
        noise_r = rr.normal(3 * nrp).reshape(nrp, 3) * np.deg2rad(0.3)
        mix = rr.normal(9 * nrp).reshape(nrp, 3, 3) * 0.1
        pb.rot_prior_frame = np.array(rot_prior_frames, np.int32)
        pb.rot_prior_q0 = np.stack([qmul(truth[f, 0:4], qexp(noise_r[k])) for k, f in enumerate(rot_prior_frames)])
        pb.rot_prior_q0 /= np.linalg.norm(pb.rot_prior_q0, axis=1, keepdims=True)
        pb.rot_prior_sqrt_info = np.stack([(np.eye(3) + mix[k]) / np.deg2rad(0.3) for k in range(nrp)]).reshape(nrp, 9)
    if duplicate_fraction > 0:
        # duplicate residual blocks (bundle_adjustor.cpp:165-179): tracks of planes with fewer than 20 members get their reprojection
        # blocks a second time (a third, when two such planes hold them).  Own random stream, like the rotation priors.
        rd = Rng(seed + 7177)
        u = rd.uniform(pb.n_landmarks)
        pb.lm_multiplicity = (1 + (u < duplicate_fraction).astype(np.int32) + (u < 0.25 * duplicate_fraction).astype(np.int32)).astype(np.int32)
    pb.frame_state = init
    pb.lm_inv_depth = rho0
    pb.truth_frame_state = truth
    pb.meta.update(dict(n_frames=N, n_landmarks=M, visibility=K, plane_tracks=n_plane, seed=seed, points=pts))
    pb._canon()
    return pb


def algorithmic_bytes_per_iteration(pb):
    """2*(20 F + 32 M) + 8 M + 8 (dN)^2 bytes per trust-region iteration."""
    d = 15 if pb.use_inertial else 6
    F, M, N = pb.n_obs, pb.n_landmarks, pb.n_frames
    return 2 * (20 * F + 32 * M) + 8 * M + 8 * (d * N) ** 2


# ---- synthetic KLT inputs ---------------------------------------------------------------------------
def make_image_pair(width=752, height=480, n_points=1500, seed=SEED, max_motion=6.0, gain=1.05, noise_sigma=2.0):
    """Band-limited random texture (64 random-phase sinusoids + noise), second image = first warped by a known
    homography (<= max_motion px) with a brightness gain.  Returns img0, img1 (u8), prev_xy, truth_xy, init_xy (float32):
    points on a jittered grid >= 20 px from the borders; init = truth + U(-2, 2) px (mimics the gyro prediction,
    frame.cpp:97-103)."""
    rng = Rng(seed + 17)
    K = 64
    fx = (rng.uniform(K) - 0.5) * 0.9
    fy = (rng.uniform(K) - 0.5) * 0.9
    ph = rng.uniform(K) * 2 * np.pi
    amp = 0.5 + rng.uniform(K)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)

    def tex(x, y):
        acc = np.zeros_like(x)
        for k in range(K):
            acc += amp[k] * np.cos(fx[k] * x + fy[k] * y + ph[k])
        return 128.0 + acc * (90.0 / np.sqrt(K))

    # homography: small rotation + translation + perspective, pixel motion bounded by max_motion
    cx, cy = width / 2.0, height / 2.0
    ang, tx, ty, p1, p2 = 0.006, 2.5, -1.8, 2e-6, -1.5e-6
    H = np.array([[np.cos(ang), -np.sin(ang), tx], [np.sin(ang), np.cos(ang), ty], [p1, p2, 1.0]])
    T = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    H = np.linalg.inv(T) @ H @ T
    Hi = np.linalg.inv(H)

    def warp(M, x, y):
        d = M[2, 0] * x + M[2, 1] * y + M[2, 2]
        return (M[0, 0] * x + M[0, 1] * y + M[0, 2]) / d, (M[1, 0] * x + M[1, 1] * y + M[1, 2]) / d

    n0 = rng.normal(width * height).reshape(height, width) * noise_sigma
    n1 = rng.normal(width * height).reshape(height, width) * noise_sigma
    img0 = np.clip(np.rint(tex(xs, ys) + n0), 0, 255).astype(np.uint8)
    sx, sy = warp(Hi, xs, ys)  # content of image 1 at (x, y) comes from image 0 at H^-1 (x, y)
    img1 = np.clip(np.rint(gain * tex(sx, sy) + n1), 0, 255).astype(np.uint8)
    # jittered grid
    gx = int(np.ceil(np.sqrt(n_points * width / height)))
    gy = int(np.ceil(n_points / gx))
    m = 30.0
    px = m + (np.arange(gx) + 0.5) * (width - 2 * m) / gx
    py = m + (np.arange(gy) + 0.5) * (height - 2 * m) / gy
    PX, PY = np.meshgrid(px, py)
    pts = np.stack([PX.ravel(), PY.ravel()], 1)[:n_points]
    pts += (rng.uniform(2 * n_points).reshape(n_points, 2) - 0.5) * 4.0
    tx_, ty_ = warp(H, pts[:, 0], pts[:, 1])
    truth = np.stack([tx_, ty_], 1)
    assert np.abs(truth - pts).max() <= max_motion
    init = truth + (rng.uniform(2 * n_points).reshape(n_points, 2) - 0.5) * 4.0
    return img0, img1, pts.astype(np.float32), truth.astype(np.float32), init.astype(np.float32)


def make_undistort_maps(width=512, height=512, k1=-0.28, k2=0.07):
    """Fixed-point remap tables (the layout pvio_hip_undistort_create takes) of a plain radial model around the image centre,
    for bench.py's ingest timing -- the real camera maps come from pvio_amd/host/undistort_maps.cpp."""
    f = 0.6 * width
    v, u = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    x, y = (u - 0.5 * width) / f, (v - 0.5 * height) / f
    r2 = x * x + y * y
    kr = 1 + (k2 * r2 + k1) * r2
    iu = np.rint((f * x * kr + 0.5 * width) * 32).astype(np.int64)
    iv = np.rint((f * y * kr + 0.5 * height) * 32).astype(np.int64)
    xy = np.stack([iu >> 5, iv >> 5], axis=-1).astype(np.int16)
    frac = ((iv & 31) * 32 + (iu & 31)).astype(np.uint16)
    return xy, frac


def permute_landmarks(pb, order):
    """The same window with its landmarks listed in another order (tests: anchor frames that are NOT sorted; the reference's block order is)."""
    import copy
    pb._canon()
    out = copy.copy(pb)
    order = np.asarray(order, np.int64)
    ptr = pb.lm_obs_ptr.astype(np.int64)
    cnt = (ptr[1:] - ptr[:-1])[order]
    out.lm_obs_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    idx = np.concatenate([np.arange(ptr[l], ptr[l + 1]) for l in order]) if len(order) else np.zeros(0, np.int64)
    out.lm_anchor_frame = pb.lm_anchor_frame[order].copy()
    out.lm_anchor_z = pb.lm_anchor_z[order].copy()
    out.obs_frame = pb.obs_frame[idx].copy()
    out.obs_z = pb.obs_z[idx].copy()
    out.lm_inv_depth = pb.lm_inv_depth[order].copy()
    if pb.lm_multiplicity is not None:
        out.lm_multiplicity = pb.lm_multiplicity[order].copy()
    if pb.truth_inv_depth is not None:
        out.truth_inv_depth = pb.truth_inv_depth[order].copy()
    out.frame_state = pb.frame_state.copy()
    return out
