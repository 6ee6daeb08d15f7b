"""Headless pipeline (pvio_amd/host/headless.*, feature_tracker.*; tools/pvio_headless.cpp is its dataset loop): a rendered
sequence -- a textured wall seen from the synthetic orbit of pvio_amd.synth, 200 Hz IMU synthesized from the same analytic
trajectory -- goes through image preprocessing, LK tracking, RANSAC, corner detection (GPU), the keyframe bootstrap, PnP,
marginalization and sliding-window bundle adjustment (GPU); the reported trajectory must follow the ground truth.

This is an end-to-end plumbing test of the rows SURVEY.md section 8 marks K6 / F3, not a parity test: there is no reference
trajectory to compare with (the reference cannot run here), so the bar is the known ground truth of the rendered scene."""
import ctypes as C

import os

import numpy as np
import pytest

import host_compare
from pvio_amd import synth

dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(dp)


W, H = 512, 384
K4 = np.array([300.0, 300.0, 255.5, 191.5])
SMALL = (352, 264, np.array([206.25, 206.25, 175.5, 131.5]))  # same field of view at 0.6875 of the size: the emulated run
TEX_PPM = 160.0  # texture pixels per metre of wall


def _texture(variant=0):
    rng = synth.Rng(synth.SEED + (17 if variant == 0 else 91))
    n = 64
    fx, fy, ph, amp = (rng.uniform(n) - 0.5) * 0.9, (rng.uniform(n) - 0.5) * 0.9, rng.uniform(n) * 2 * np.pi, 0.5 + rng.uniform(n)

    def tex(x, y):
        acc = np.zeros_like(x)
        for k in range(n):
            acc += amp[k] * np.cos(fx[k] * x + fy[k] * y + ph[k])
        return 128.0 + acc * (90.0 / np.sqrt(n))
    return tex


def _relief(u, v, variant=0):
    """height [m] of the relief scene above the wall plane at wall coordinates (u, v) [m]: smooth, +-0.45 m over ~1.5 m -- no patch of it large enough to
    hold 30 tracks lies within the 3 cm of the reference's plane RANSAC (core/plane_extractor.cpp:55-57), so PlaneExtractor never reports a plane.
    variant 1 (scene "..._b", round 6: a second family of long sequences): other wavelengths and phases"""
    if variant == 1:
        return 0.22 * np.sin(3.0 * u - 1.1) * np.cos(3.6 * v + 0.5) + 0.2 * np.sin(1.7 * u + 2.4 * v - 0.4)
    return 0.25 * np.sin(4.1 * u + 0.3) * np.cos(3.3 * v - 0.8) + 0.2 * np.sin(2.2 * u - 1.9 * v + 1.1)


def _undistorted_rays(us, vs, K4, dist, model="radtan"):
    """normalized pinhole coordinates (x, y) of the scene point seen at DISTORTED pixel (u, v) under OpenCV's radial-tangential model
    (k1, k2, p1, p2): the inverse of x_d = x (1 + k1 r2 + k2 r4) + 2 p1 x y + p2 (r2 + 2 x^2), ... by fixed-point iteration (cv::undistortPoints);
    model "equidistant" (Kannala-Brandt, image_undistorter.h:78-93): theta_d = theta (1 + k1 theta^2 + ... + k4 theta^8) inverted by Newton steps"""
    xd, yd = (us - K4[2]) / K4[0], (vs - K4[3]) / K4[1]
    if model == "equidistant":
        k1, k2, k3, k4 = dist
        rd = np.sqrt(xd * xd + yd * yd)
        th = rd.copy()
        for _ in range(10):
            t2 = th * th
            f = th * (1 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) - rd
            df = 1 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)))
            th = th - f / df
        th = np.minimum(th, 1.45)  # rays beyond ~83 degrees see nothing sensible of the wall
        sc = np.where(rd > 1e-12, np.tan(th) / np.maximum(rd, 1e-12), 1.0)
        return xd * sc, yd * sc
    k1, k2, p1, p2 = dist
    x, y = xd.copy(), yd.copy()
    for _ in range(12):
        r2 = x * x + y * y
        rad = 1.0 + k1 * r2 + k2 * r2 * r2
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        x, y = (xd - dx) / rad, (yd - dy) / rad
    return x, y


# A trajectory for LONG sequences: synth._pose orbits the centre at a constant rate and leaves the +-40 degrees in front of the wall after
# ~4 s; this one sweeps back and forth on the same circle -- theta(t) = TH0 + A sin(w t), peak rate A w = OMEGA as before -- so that any number of frames
# keeps the wall (or the relief) in view.  Same frame conventions, analytic velocity / acceleration / yaw rate for the IMU.
SWEEP_A, SWEEP_W = 0.6, 0.5
SWEEP_B = (0.42, 0.85, 1.6)  # variant 1: a narrower, faster sweep (peak rate 0.36 rad/s, a reversal every 3.7 s) with a faster, larger height oscillation


def _pose_sweep(t, variant=0):
    R0, H_AMP, H_FREQ = synth.RADIUS, synth.H_AMP, synth.H_FREQ
    A, Wf = (SWEEP_A, SWEEP_W) if variant == 0 else SWEEP_B[:2]
    if variant == 1:
        H_AMP, H_FREQ = 1.3 * H_AMP, SWEEP_B[2] * H_FREQ
    th, dth, ddth = A * np.sin(Wf * t), A * Wf * np.cos(Wf * t), -A * Wf ** 2 * np.sin(Wf * t)
    c, s = np.cos(th), np.sin(th)
    p = np.array([R0 * c, R0 * s, H_AMP * np.sin(H_FREQ * t)])
    v = np.array([-R0 * s * dth, R0 * c * dth, H_AMP * H_FREQ * np.cos(H_FREQ * t)])
    acc = np.array([-R0 * (c * dth ** 2 + s * ddth), R0 * (-s * dth ** 2 + c * ddth), -H_AMP * H_FREQ ** 2 * np.sin(H_FREQ * t)])
    up, fwd = np.array([0.0, 0.0, 1.0]), np.array([-c, -s, 0.0])
    R = np.stack([up, np.cross(fwd, up), fwd], 1)
    return R, p, v, acc, dth


def render_sequence(n_frames, fps=20.0, imu_rate=200.0, t_start=0.0, size=None, relief=False, distortion=None, model="radtan", extrinsics=None, sweep=False, variant=0):
    """images (n, H, W) u8, image times, IMU (t, w, a), body poses at the image times (t p q).  relief: the textured surface is the wall plus _relief()
    (ray / surface intersection by fixed-point iteration) instead of the wall itself: a scene without planes."""
    W, H, K4 = size if size is not None else (globals()["W"], globals()["H"], globals()["K4"])
    q_bc = synth.Q_BC / np.linalg.norm(synth.Q_BC)
    R_bc, p_bc = synth.qmat(q_bc), synth.P_BC
    # the wall: through the orbit centre, facing the camera at mid-sequence, tilted by 20 degrees so that depth varies
    pose = (lambda t: _pose_sweep(t, variant)[:4]) if sweep else synth._pose
    yaw_rate = (lambda t: _pose_sweep(t, variant)[4]) if sweep else (lambda t: synth.OMEGA)
    t_mid = 0.0 if sweep else t_start + 0.5 * n_frames / fps  # (sweep: the wall faces the centre of the sweep)
    R_mid, p_mid, _, _ = pose(t_mid)
    fwd = R_mid[:, 2]
    tilt = np.deg2rad(20.0)
    nrm = np.cos(tilt) * fwd + np.sin(tilt) * R_mid[:, 1]
    d = 0.0  # n . X = 0: the plane passes through the orbit centre (the origin)
    if extrinsics is not None:  # another rig (q_bc xyzw, p_bc): the wall is set up against THAT camera's axes, 3 m in front of it at mid-sequence
        q_bc = np.asarray(extrinsics[0], float) / np.linalg.norm(extrinsics[0])
        R_bc, p_bc = synth.qmat(q_bc), np.asarray(extrinsics[1], float)
        Rc = R_mid @ R_bc
        fwd = Rc[:, 2]
        nrm = np.cos(tilt) * fwd + np.sin(tilt) * Rc[:, 0]
        d = float(nrm @ (p_mid + R_mid @ p_bc + 3.0 * fwd)) / np.linalg.norm(nrm)
    nrm /= np.linalg.norm(nrm)
    e1 = np.cross(nrm, np.array([0.0, 0.0, 1.0]))
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(nrm, e1)
    tex = _texture(variant)
    us, vs = np.meshgrid(np.arange(W, dtype=float), np.arange(H, dtype=float))
    if distortion is None:
        rays_c = np.stack([(us - K4[2]) / K4[0], (vs - K4[3]) / K4[1], np.ones_like(us)], -1)
    else:  # the images a camera with this lens records: a reader that undistorts them (cv::undistort with the same K) recovers the pinhole image
        xn, yn = _undistorted_rays(us, vs, K4, distortion, model)
        rays_c = np.stack([xn, yn, np.ones_like(us)], -1)
    rng = np.random.default_rng(5)
    images, times, poses = [], [], []
    for k in range(n_frames):
        t = t_start + k / fps
        R, p, _, _ = pose(t)
        R_wc, p_wc = R @ R_bc, p + R @ p_bc
        rays = rays_c @ R_wc.T
        s = (d - nrm @ p_wc) / (rays @ nrm)
        X = p_wc + rays * s[..., None]
        if relief:  # n . X = d + h(u, v): six fixed-point steps (the rays are within 40 degrees of the normal, |grad h| < 1.5)
            for _ in range(6):
                s = (d + _relief(X @ e1, X @ e2, variant) - nrm @ p_wc) / (rays @ nrm)
                X = p_wc + rays * s[..., None]
        img = tex(TEX_PPM * (X @ e1), TEX_PPM * (X @ e2)) + rng.normal(0, 1.5, (H, W))
        images.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
        times.append(t)
        poses.append(np.concatenate([[t], p, synth.mat2q(R)]))
    n_imu = int(round((n_frames / fps + 0.1) * imu_rate))
    imu_t = t_start - 0.02 + np.arange(n_imu) / imu_rate
    imu_w, imu_a = np.zeros((n_imu, 3)), np.zeros((n_imu, 3))
    bg, ba = np.full(3, 2e-3), np.full(3, 2e-2)
    nw = rng.normal(0, np.sqrt(synth.COV_G * imu_rate), (n_imu, 3))
    na = rng.normal(0, np.sqrt(synth.COV_A * imu_rate), (n_imu, 3))
    for k in range(n_imu):
        R, _, _, acc = pose(imu_t[k])
        imu_w[k] = np.array([yaw_rate(imu_t[k]), 0.0, 0.0]) + bg + nw[k]  # body x = world z (synth._pose)
        imu_a[k] = R.T @ (acc + np.array([0, 0, synth.GRAVITY])) + ba + na[k]
    return (np.ascontiguousarray(np.stack(images)), np.array(times), imu_t, np.ascontiguousarray(imu_w), np.ascontiguousarray(imu_a),
            np.ascontiguousarray(np.stack(poses)), q_bc, p_bc)


def run(lib, n_frames, window, gap, distance=25.0, size=None):
    W, H, K4 = size if size is not None else (globals()["W"], globals()["H"], globals()["K4"])
    import os
    # every keyframe solve also re-walks the Map from scratch and compares with the incrementally flattened window
    os.environ["PVIO_HIP_FLATTEN_VERIFY"] = "1"
    images, times, imu_t, imu_w, imu_a, gt, q_bc, p_bc = render_sequence(n_frames, size=size)
    out, stats = np.zeros((n_frames, 8)), np.zeros(4, np.int32)
    err = C.create_string_buffer(512)
    lib.host_headless_run.restype = C.c_int
    rc = lib.host_headless_run(C.c_int(n_frames), C.c_int(W), C.c_int(H), images.ctypes.data_as(C.POINTER(C.c_uint8)), _d(times), C.c_int(len(imu_t)),
                               _d(imu_t), _d(imu_w), _d(imu_a), _d(K4), _d(np.ascontiguousarray(q_bc)), _d(np.ascontiguousarray(p_bc)), C.c_int(len(gt)),
                               _d(gt), C.c_int(window), C.c_int(gap), C.c_double(distance), _d(out), stats.ctypes.data_as(C.POINTER(C.c_int32)), err,
                               C.c_int(512))
    assert rc == 0, err.value.decode()
    return out, stats, gt


@pytest.mark.gpu
def test_headless_pipeline_follows_the_rendered_trajectory():
    lib = host_compare.load("libpvio_host.so")
    n_frames, window, gap = 60, 6, 3
    out, stats, gt = run(lib, n_frames, window, gap)
    assert stats[0] == 1, "the window was never bootstrapped"
    assert stats[1] == window + 1 or stats[1] == window  # steady state: sliding_window_size (+ 1 while a non-keyframe rides on top)
    assert stats[2] >= 3 and stats[3] >= 40              # keyframe solves after the bootstrap; a populated window
    valid = np.abs(out[:, 4:8]).sum(1) > 0
    first = int(np.argmax(valid))
    assert first <= (window - 1) * gap + 3 and valid[first:].all()
    err = np.linalg.norm(out[valid, 1:4] - gt[valid, 1:4], axis=1)
    # output poses are IMU-propagated from the newest optimized frame (core.cpp:142-163): a few centimetres on a 3 m orbit
    print("headless: first pose at frame", first, "solves", stats[2], "tracks", stats[3], "position error cm: median %.2f max %.2f" % (100 * np.median(err), 100 * err.max()))
    assert np.median(err) < 0.05 and err.max() < 0.15
    qe = np.abs(np.sum(out[valid, 4:8] * gt[valid, 4:8], axis=1))
    assert np.degrees(2 * np.arccos(np.clip(qe, 0, 1))).max() < 2.0


# the emulated run of the same pipeline (no GPU) is tests/test_chain_parity.py::test_chain_parity_emulated: 12 frames through the
# kernel emulator, compared record by record with the oracle chain and with the ground truth


def _binary_on_dataset_layout(tmp_path, kind, binary, n_frames=36, window="6", gap="3", timeout=600, bounds=(0.05, 0.12), min_poses=12):
    """writes a rendered sequence to disk in the dataset's layout, runs `binary` (a build of tools/pvio_headless.cpp) on it, checks trajectory.tum"""
    import subprocess
    import test_host_ingest as ing
    if kind == "euroc":
        W_, H_, K4e, dist, model, ext = 752, 480, np.array([458.654, 457.296, 367.215, 248.375]), ing.EUROC_D, "radtan", None
    else:  # config/tum-vi.yaml:13-22
        W_, H_, K4e, dist, model = 512, 512, np.array([ing.TUM_K[0], ing.TUM_K[4], ing.TUM_K[2], ing.TUM_K[5]]), ing.TUM_D, "equidistant"
        ext = ([-0.013272, -0.694726, 0.719112, 0.007648], [0.04536566, -0.071996, -0.04478181])
    images, times, imu_t, imu_w, imu_a, gt, q_bc, p_bc = render_sequence(n_frames, size=(W_, H_, K4e), relief=True, distortion=dist, model=model, extrinsics=ext)
    t0 = 20_000_000_000  # ns
    cam_ns = [t0 + int(round(t * 1e9)) for t in times]
    imu_rows = [(t0 + int(round(t * 1e9)), *w, *a) for t, w, a in zip(imu_t, imu_w, imu_a)]
    root = tmp_path / "mav0"
    ing._write_sequence(root, kind == "euroc", list(images), cam_ns, imu_rows)
    gt_path, out_path = tmp_path / "gt.tum", tmp_path / "trajectory.tum"
    with open(gt_path, "w") as f:
        for g in gt:
            f.write(" ".join(repr(float(v)) for v in ([g[0] + t0 * 1e-9] + list(g[1:]))) + "\n")
    r = subprocess.run([binary, kind + "://" + str(root), str(gt_path), str(out_path), "-1", window, gap], capture_output=True, text=True,
                       timeout=timeout)  # window of 6 keyframes 3 frames apart: the first window exists at frame 15 of the 36
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l.split() for l in open(out_path).read().strip().splitlines()]
    assert len(lines) >= min_poses and all(len(l) == 8 for l in lines), (len(lines), r.stderr[-500:])
    assert all("." in l[1] and len(l[1].split(".")[1]) >= 9 for l in lines)  # full precision, not %g
    tum = np.array(lines, float)
    idx = [int(np.argmin(np.abs(gt[:, 0] + t0 * 1e-9 - t))) for t in tum[:, 0]]
    assert np.abs(gt[idx, 0] + t0 * 1e-9 - tum[:, 0]).max() < 1e-6
    err = np.linalg.norm(tum[:, 1:4] - gt[idx, 1:4], axis=1)
    print(os.path.basename(binary) + " on the " + kind + "-layout sequence: %d poses, position error cm: median %.2f max %.2f; %s" % (
        len(tum), 100 * np.median(err), 100 * err.max(), r.stderr.strip().splitlines()[-1]))
    assert np.median(err) < bounds[0] and err.max() < bounds[1]
    return tum


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["euroc", "tum"])
def test_headless_binary_on_a_dataset_layout_sequence(tmp_path, kind):
    """BASELINE configs[0]'s plumbing with the actual binary (tools/pvio_headless.cpp, the sequence loop of pvio-pc/src/main.cpp:207-258): a sequence ON DISK in
    EuRoC's layout -- cam0/data.csv + cam0/data/<ns>.png recorded through EuRoC's lens (752 x 480, the intrinsics and radial-tangential coefficients of
    euroc_dataset_reader.cpp:73-74), imu0/data.csv at 200 Hz, CRLF line ends -- goes through the readers (CSV, PNG decode), the device undistortion, the front
    end, PnP and the sliding-window BA, and comes out as trajectory.tum (output_writer.h:41-50 format), which follows the ground truth.
    kind "tum": TUM-VI's layout and camera (512 x 512 equidistant fisheye, tum_dataset_reader.cpp:73-81, LF line ends).
    This is the build above the STAND-IN control plane (tests/host/standin); the one above the reference's own pvio::PVIO is the next test."""
    import subprocess
    host_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
    subprocess.check_call(["make", "-s", "-C", host_dir, "pvio_headless"])
    _binary_on_dataset_layout(tmp_path, kind, os.path.join(host_dir, "pvio_headless"))


def _reference_pvio_binary(name):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle", "ref"), "headless"])
    path = os.path.join(root, "oracle", "_ref", name)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/%s not built and /root/reference absent" % name)
    return path


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["euroc", "tum"])
def test_headless_binary_above_the_reference_pvio(tmp_path, kind):
    """the SAME tools/pvio_headless.cpp built against the reference's own pvio::PVIO (oracle/ref/Makefile `headless`: the reference's
    pvio.cpp, core/*.cpp, map/*.cpp compiled from /root/reference, the product's BundleAdjustor / visual_inertial_pnp / HipImage / readers linked in,
    nothing from tests/) on the same on-disk sequences: PNG + CSV readers -> device undistortion -> PVIO::track_* -> trajectory.tum."""
    _binary_on_dataset_layout(tmp_path, kind, _reference_pvio_binary("pvio_headless"))


def test_headless_binary_above_the_reference_pvio_emulated(tmp_path):
    """the same binary above the kernel emulator (CPU suite): 14 frames of the TUM-VI-layout sequence through PNG / CSV readers, emulated undistortion, front end,
    the reference's pvio::PVIO, PnP and the window solves; a window of 3 keyframes 2 frames apart exists from frame 6 on"""
    _binary_on_dataset_layout(tmp_path, "tum", _reference_pvio_binary("pvio_headless_emu"), n_frames=14, window="3", gap="2", timeout=1200, bounds=(0.06, 0.12), min_poses=6)
