"""Host side of the FeatureTracker seam (pvio_amd/host/feature_front.*): Poisson-disk filter, survivor selection, gyro
keypoint prediction, and the pvio::Image adapter over the C ABI.  Checked against the independent restatement in
oracle/oracle_front.cpp and against brute-force properties (the reference has no tests for this code)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import host_compare
from oracle import oracle_py
from pvio_amd import synth

u8p, f64p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.POINTER(C.c_uint64)


def _p(a, t):
    return a.ctypes.data_as(t)


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"), "libpvio_hipemu.so"])
    return host_compare.load("libpvio_host_emu.so")


@pytest.fixture(scope="module")
def ora():
    oracle_py.build()
    return oracle_py.lib()


def poisson(lib, prefix, radius, preset, pts):
    acc = np.zeros(len(pts), np.uint8)
    preset = np.ascontiguousarray(preset, np.float64).reshape(-1, 2)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    getattr(lib, prefix + "_poisson_insert")(C.c_double(radius), C.c_int(len(preset)), _p(preset if len(preset) else np.zeros((1, 2)), f64p),
                                             C.c_int(len(pts)), _p(pts, f64p), _p(acc, u8p))
    return acc.astype(bool)


@pytest.mark.parametrize("seed,radius,n,extent", [(1, 20.0, 400, 300.0), (2, 7.5, 1500, 200.0), (3, 20.0, 50, 2000.0), (4, 1.0, 300, 6.0)])
def test_poisson_filter_matches_oracle_and_is_a_maximal_packing(host, ora, seed, radius, n, extent):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-extent / 3, extent, (n, 2))  # negative coordinates too: the cell index is floor(), not truncation
    a_host, a_ora = poisson(host, "host", radius, [], pts), poisson(ora, "oracle", radius, [], pts)
    assert (a_host == a_ora).all()
    acc = pts[a_host]
    d = np.linalg.norm(acc[:, None, :] - acc[None, :, :], axis=2) + np.eye(len(acc)) * 1e9
    assert d.min() >= radius  # the skipped corner cell and the extra probed cell never matter geometrically
    for i in np.nonzero(~a_host)[0]:  # greedy in order: a rejected candidate is within radius of an EARLIER accepted one
        earlier = pts[:i][a_host[:i]]
        assert (np.linalg.norm(earlier - pts[i], axis=1) < radius).any()


def test_poisson_preset_points_block_and_overwrite_their_cell(host, ora):
    rng = np.random.default_rng(7)
    preset = rng.uniform(0, 100, (40, 2))
    pts = rng.uniform(0, 100, (300, 2))
    a_host, a_ora = poisson(host, "host", 10.0, preset, pts), poisson(ora, "oracle", 10.0, preset, pts)
    assert (a_host == a_ora).all()
    # two presets in one cell: only the later one is remembered, so a candidate next to the EARLIER one can pass
    r = 10.0
    cell = r / np.sqrt(2.0)
    first, second = np.array([0.1 * cell, 0.1 * cell]), np.array([0.9 * cell, 0.9 * cell])
    cand = np.array([[-0.5 * cell, 0.1 * cell]])  # 0.6 cell from `first` (< r), farther than r from `second`
    assert np.linalg.norm(cand[0] - first) < r <= np.linalg.norm(cand[0] - second)
    for lib, prefix in ((host, "host"), (ora, "oracle")):
        assert poisson(lib, prefix, r, [first], cand)[0] == False  # noqa: E712
        assert poisson(lib, prefix, r, [first, second], cand)[0] == True  # noqa: E712


@pytest.mark.parametrize("seed,n,ties", [(11, 300, False), (12, 800, True), (13, 5, True), (14, 0, False)])
def test_survivor_selection_matches_oracle(host, ora, seed, n, ties):
    rng = np.random.default_rng(seed)
    nxt = np.ascontiguousarray(rng.uniform(20, 500, (max(n, 1), 2)))
    length = (rng.integers(0, 4 if ties else 10**6, max(n, 1))).astype(np.uint64)  # 0 = keypoint without a track
    status0 = (rng.uniform(size=max(n, 1)) < 0.8).astype(np.uint8)
    out = {}
    for lib, prefix in ((host, "host"), (ora, "oracle")):
        st = status0.copy()
        getattr(lib, prefix + "_select_tracked")(C.c_int(n), _p(nxt, f64p), _p(length, u64p), C.c_double(20.0), _p(st, u8p))
        out[prefix] = st
    assert (out["host"] == out["oracle"]).all()
    st = out["host"][:n].astype(bool)
    keep = st & (length[:n] > 0)
    assert not (st & ~status0[:n].astype(bool)).any()  # nothing is resurrected
    assert (st[length[:n] == 0] == status0[:n][length[:n] == 0].astype(bool)).all()  # trackless keypoints are left alone
    if keep.sum() > 1:
        p = nxt[:n][keep]
        d = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(len(p)) * 1e9
        assert d.min() >= 20.0
    if not ties and n:  # with distinct lengths the longest surviving input always survives
        cand = np.nonzero(status0[:n].astype(bool) & (length[:n] > 0))[0]
        if len(cand):
            assert st[cand[np.argmax(length[:n][cand])]]


def _rand_q(rng, angle):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    return np.concatenate([ax * np.sin(angle / 2), [np.cos(angle / 2)]])


def test_keypoint_prediction_matches_oracle_and_rotates_bearings(host, ora):
    rng = np.random.default_rng(5)
    kp = np.ascontiguousarray(rng.uniform(-0.6, 0.6, (200, 2)))
    K = np.array([458.654, 457.296, 367.215, 248.375])
    q_ci, q_ii, dq, q_ij, q_cj = (_rand_q(rng, a) for a in (0.3, 0.0, 0.05, 0.0, 0.3))
    q_cj = q_ci.copy()  # same rig
    out = {}
    for lib, prefix in ((host, "host"), (ora, "oracle")):
        o = np.zeros_like(kp)
        getattr(lib, prefix + "_predict_keypoints")(_p(q_ci, f64p), _p(q_ii, f64p), _p(dq, f64p), _p(q_ij, f64p), _p(q_cj, f64p), _p(K, f64p),
                                                    C.c_int(len(kp)), _p(kp, f64p), _p(o, f64p))
        out[prefix] = o
    assert np.abs(out["host"] - out["oracle"]).max() < 1e-9
    # no rotation between the frames: the prediction is the keypoint itself, in pixels
    ident = np.array([0.0, 0, 0, 1])
    o = np.zeros_like(kp)
    host.host_predict_keypoints(_p(q_ci, f64p), _p(q_ii, f64p), _p(ident, f64p), _p(q_ij, f64p), _p(q_cj, f64p), _p(K, f64p), C.c_int(len(kp)), _p(kp, f64p), _p(o, f64p))
    assert np.abs(o - (kp * K[:2] + K[2:])).max() < 1e-9
    # 3 degrees of yaw move the image centre by about f * tan(3 deg)
    assert 15 < np.abs(out["host"] - (kp * K[:2] + K[2:])).max() < 60


# ---- the same three pieces against the REFERENCE'S OWN code: Frame::track_keypoints (map/frame.cpp:89-139) and
# PoissonDiskFilter<2> (utility/poisson_disk_filter.h:25-130) compiled unedited into oracle/_ref/libpvio_ref.so, driven through a scripted
# pvio::Image (oracle/ref/ref_capi.cpp: ref_poisson_insert / ref_select_tracked / ref_predict_keypoints, same signatures) --------------

@pytest.fixture(scope="module")
def refl():
    from oracle import ref_py
    if not ref_py.available():
        pytest.skip("oracle/_ref/libpvio_ref.so not built and /root/reference absent")
    return ref_py.lib()


@pytest.mark.parametrize("seed,radius,n,extent,n_preset", [(1, 20.0, 400, 300.0, 0), (2, 7.5, 1500, 200.0, 0), (3, 20.0, 50, 2000.0, 0), (4, 1.0, 300, 6.0, 0), (5, 10.0, 300, 100.0, 40),
                                                           (6, 25.0, 1000, 752.0, 150)])
def test_poisson_filter_matches_reference_source(host, ora, refl, seed, radius, n, extent, n_preset):
    """product PoissonDisk2 == oracle restatement == the reference's PoissonDiskFilter<2> (its probe-order quirk included), candidate by candidate"""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-extent / 3, extent, (n, 2))
    preset = rng.uniform(0, extent, (n_preset, 2))
    a_ref = poisson(refl, "ref", radius, preset, pts)
    assert (poisson(host, "host", radius, preset, pts) == a_ref).all()
    assert (poisson(ora, "oracle", radius, preset, pts) == a_ref).all()
    assert 0 < a_ref.sum() < n
    # the overwrite quirk of preset_point (:40-44), on the reference itself
    r = 10.0
    cell = r / np.sqrt(2.0)
    first, second = np.array([0.1 * cell, 0.1 * cell]), np.array([0.9 * cell, 0.9 * cell])
    cand = np.array([[-0.5 * cell, 0.1 * cell]])
    assert poisson(refl, "ref", r, [first], cand)[0] == False  # noqa: E712
    assert poisson(refl, "ref", r, [first, second], cand)[0] == True  # noqa: E712


@pytest.mark.parametrize("seed,n,max_len", [(11, 300, 40), (12, 800, 4), (13, 5, 3), (15, 1500, 12), (16, 200, 2)])
def test_survivor_selection_matches_reference_source(host, ora, refl, seed, n, max_len):
    """Frame::track_keypoints' survivor selection (:108-130): LK survivors ordered by track length (std::sort: ties in the library's order), accepted by
    the Poisson-disk filter in that order.  The reference runs on real Frame / Track objects with tracks of the given lengths."""
    rng = np.random.default_rng(seed)
    nxt = np.ascontiguousarray(rng.uniform(20, 500, (n, 2)))
    length = rng.integers(0, max_len + 1, n).astype(np.uint64)  # 0 = keypoint without a track
    status0 = (rng.uniform(size=n) < 0.8).astype(np.uint8)
    out = {}
    for lib, prefix in ((host, "host"), (ora, "oracle"), (refl, "ref")):
        st = status0.copy()
        getattr(lib, prefix + "_select_tracked")(C.c_int(n), _p(nxt, f64p), _p(length, u64p), C.c_double(20.0), _p(st, u8p))
        out[prefix] = st
    assert (out["host"] == out["ref"]).all(), np.nonzero(out["host"] != out["ref"])
    assert (out["oracle"] == out["ref"]).all()
    assert 0 < out["ref"].sum() <= status0.sum() and (n < 100 or out["ref"].sum() < status0.sum())


def test_keypoint_prediction_matches_reference_source(host, ora, refl):
    """Frame::track_keypoints' gyro-only prediction (:97-103): what the reference hands to Image::track_keypoints as the initial guess"""
    rng = np.random.default_rng(5)
    kp = np.ascontiguousarray(rng.uniform(-0.6, 0.6, (200, 2)))
    K = np.array([458.654, 457.296, 367.215, 248.375])
    for trial in range(5):
        q_ci, q_ii, dq, q_ij, q_cj = (_rand_q(rng, a) for a in (0.3, 0.1 * trial, 0.05 * (trial + 1), 0.1 * trial, 0.3))
        out = {}
        for lib, prefix in ((host, "host"), (ora, "oracle"), (refl, "ref")):
            o = np.zeros_like(kp)
            getattr(lib, prefix + "_predict_keypoints")(_p(q_ci, f64p), _p(q_ii, f64p), _p(dq, f64p), _p(q_ij, f64p), _p(q_cj, f64p), _p(K, f64p),
                                                        C.c_int(len(kp)), _p(kp, f64p), _p(o, f64p))
            out[prefix] = o
        assert np.abs(out["host"] - out["ref"]).max() < 1e-9
        assert np.abs(out["oracle"] - out["ref"]).max() < 1e-9
        assert np.abs(out["ref"] - (kp * K[:2] + K[2:])).max() > 1.0


def _image_seam(lib, oracle, w, h, n):
    img0, img1, p, truth, init = synth.make_image_pair(w, h, n)
    P0, P1 = oracle.build_pyramid(oracle.clahe(img0)), oracle.build_pyramid(oracle.clahe(img1))
    n_ref, s_ref = oracle.klt_track(P0, P1, p, init)
    gate = (n_ref[:, 0] < 20) | (n_ref[:, 0] >= w - 20) | (n_ref[:, 1] < 20) | (n_ref[:, 1] >= h - 20)  # opencv_image.cpp:106-108
    s_ref = np.where(gate, 0, s_ref)
    cur = np.ascontiguousarray(p, np.float64)
    nxt = np.ascontiguousarray(init, np.float64)
    st = np.zeros(n, np.uint8)
    err = C.create_string_buffer(256)
    rc = lib.host_image_track(_p(np.ascontiguousarray(img0), u8p), _p(np.ascontiguousarray(img1), u8p), C.c_int(w), C.c_int(h), C.c_int(n), _p(cur, f64p),
                              _p(nxt, f64p), C.c_int(1), C.c_int(0), _p(st, u8p), err, C.c_int(256))
    assert rc == 0, err.value
    assert (st.astype(bool) == (s_ref > 0)).all()
    ok = st > 0
    assert np.abs(nxt - n_ref)[ok].max() == 0.0  # bit-identical: the kernel sums in the oracle's defined order (oracle_klt.cpp header)
    if (~ok).any():
        assert np.abs(nxt - init)[~ok].max() == 0.0  # failed tracks keep the caller's value (opencv_image.cpp:131-136)
    return int(ok.sum())


def test_image_seam_emulated(host, oracle):
    assert _image_seam(host, oracle, 160, 120, 48) > 20


@pytest.mark.gpu
def test_image_seam_gpu(oracle):
    lib = host_compare.load("libpvio_host.so")
    assert _image_seam(lib, oracle, 512, 512, 1200) > 900


def _detect_seam(lib, oracle, w, h):
    """pvio::Image::detect_keypoints through the host adapter == oracle response map + goodFeaturesToTrack selection, then
    the reference's own post-processing (Poisson filter against the existing points, 20 px border) done independently here."""
    img0, _, _, _, _ = synth.make_image_pair(w, h, 8)
    r = oracle.harris_response(oracle.clahe(img0))
    xy, resp = oracle.good_features(r, 1000, 1e-3, 20.0)
    rng = np.random.default_rng(9)
    existing = np.ascontiguousarray(np.stack([rng.uniform(20, w - 20, 12), rng.uniform(20, h - 20, 12)], 1))
    dist = 15.0
    acc = poisson(oracle_lib(), "oracle", dist, existing, xy.astype(np.float64))
    want = [tuple(p) for p, a in zip(xy.astype(np.float64), acc) if a and not (p[0] < 20 or p[1] < 20 or p[0] >= w - 20 or p[1] >= h - 20)]
    out = np.zeros((2000, 2))
    err = C.create_string_buffer(256)
    lib.host_image_detect.restype = C.c_int
    n = lib.host_image_detect(_p(np.ascontiguousarray(img0), u8p), C.c_int(w), C.c_int(h), C.c_int(len(existing)), _p(existing, f64p), C.c_double(dist), C.c_int(2000),
                              _p(out, f64p), err, C.c_int(256))
    assert n >= 0, err.value
    got = out[:n]
    assert (got[:len(existing)] == existing).all()  # existing keypoints stay where they are, new ones are appended
    assert [tuple(p) for p in got[len(existing):]] == want
    return len(want)


def oracle_lib():
    oracle_py.build()
    return oracle_py.lib()


def test_detect_seam_emulated(host, oracle):
    assert _detect_seam(host, oracle, 200, 160) > 5


@pytest.mark.gpu
def test_detect_seam_gpu(oracle):
    lib = host_compare.load("libpvio_host.so")
    assert _detect_seam(lib, oracle, 752, 480) > 100


def test_image_seam_with_ransac_rejects_planted_outliers(host, oracle):
    """Full track_keypoints semantics (LK + border gate + fundamental-matrix RANSAC): tracks whose initial guess is far off
    converge somewhere else and are inconsistent with the epipolar geometry of the rest -> dropped by the RANSAC stage."""
    w, h, n = 240, 200, 160
    img0, img1, p, truth, init = synth.make_image_pair(w, h, n)
    cur, nxt0 = np.ascontiguousarray(p, np.float64), np.ascontiguousarray(init, np.float64)
    res = {}
    for use_ransac in (0, 1):
        nxt, st, err = nxt0.copy(), np.zeros(n, np.uint8), C.create_string_buffer(256)
        rc = host.host_image_track(_p(np.ascontiguousarray(img0), u8p), _p(np.ascontiguousarray(img1), u8p), C.c_int(w), C.c_int(h), C.c_int(n), _p(cur, f64p),
                                   _p(nxt, f64p), C.c_int(1), C.c_int(use_ransac), _p(st, u8p), err, C.c_int(256))
        assert rc == 0, err.value
        res[use_ransac] = (st.astype(bool), nxt)
    lk, (rs, nxt) = res[0][0], res[1]
    assert not (rs & ~lk).any() and rs.sum() >= 8  # RANSAC only removes
    e = np.linalg.norm(nxt - truth, axis=1)
    assert np.median(e[rs]) < 0.3
    # what it removed is, on average, worse than what it kept
    if (lk & ~rs).any():
        assert e[lk & ~rs].mean() >= e[rs].mean()


def test_image_evaluate_is_the_bicubic_sample_of_the_preprocessed_level(host, oracle):
    """pvio::Image::evaluate(u[, ddu], level) (pvio.h:125-126; opencv_image.cpp:36-52 = ceres::BiCubicInterpolator over the level's
    pixels, Catmull-Rom, indices clamped to the image, level scale = 1 / integer quotient of the extents): on integer pixel
    positions of level 0 it returns the CLAHE'd pixel itself, in between it equals a numpy restatement of the spline, and the
    gradient matches central differences of the value."""
    from pvio_amd import synth
    w, h = 160, 128
    img = synth.make_image_pair(w, h, 8)[0]
    levels = oracle.build_pyramid(oracle.clahe(img))
    rng = np.random.default_rng(3)

    def run(level, uv):
        uv = np.ascontiguousarray(uv, float)
        val, grad = np.zeros(len(uv)), np.zeros((len(uv), 2))
        err = C.create_string_buffer(256)
        host.host_image_evaluate.restype = C.c_int
        rc = host.host_image_evaluate(_p(np.ascontiguousarray(img), u8p), C.c_int(w), C.c_int(h), C.c_int(level), C.c_int(len(uv)), _p(uv, f64p), _p(val, f64p),
                                      _p(grad, f64p), err, C.c_int(256))
        assert rc == 0, err.value.decode()
        return val, grad

    def cubic(p0, p1, p2, p3, x):
        a, b, c = 0.5 * (-p0 + 3 * p1 - 3 * p2 + p3), 0.5 * (2 * p0 - 5 * p1 + 4 * p2 - p3), 0.5 * (-p0 + p2)
        return p1 + x * (c + x * (b + x * a))

    def ref(level, u):
        L = levels[level][0].astype(float) if isinstance(levels[level], (tuple, list)) else levels[level].astype(float)
        hh, ww = L.shape
        sx, sy = 1.0 / ((w - 1) // (ww - 1)), 1.0 / ((h - 1) // (hh - 1))
        c, r = u[0] * sx, u[1] * sy
        col, row = int(np.floor(c)), int(np.floor(r))
        px = lambda rr, cc: L[min(max(rr, 0), hh - 1), min(max(cc, 0), ww - 1)]
        f = [cubic(px(row - 1 + k, col - 1), px(row - 1 + k, col), px(row - 1 + k, col + 1), px(row - 1 + k, col + 2), c - col) for k in range(4)]
        return cubic(f[0], f[1], f[2], f[3], r - row)

    ints = np.column_stack([rng.integers(0, w, 20), rng.integers(0, h, 20)]).astype(float)
    v, _ = run(0, ints)
    L0 = levels[0][0] if isinstance(levels[0], (tuple, list)) else levels[0]
    assert (v == L0[ints[:, 1].astype(int), ints[:, 0].astype(int)]).all()
    for level in (0, 2):
        uv = np.column_stack([rng.uniform(-2, w + 1, 40), rng.uniform(-2, h + 1, 40)])
        v, g = run(level, uv)
        np.testing.assert_allclose(v, [ref(level, u) for u in uv], rtol=0, atol=1e-9)
        eps = 1e-5
        inner = (np.abs(uv - np.rint(uv)) > 1e-3).all(axis=1)  # the spline's derivative jumps at the knots
        for k in range(2):
            d = np.zeros(2)
            d[k] = eps
            fd = (np.array([ref(level, u + d) for u in uv]) - np.array([ref(level, u - d) for u in uv])) / (2 * eps)
            np.testing.assert_allclose(g[inner, k], fd[inner], rtol=1e-5, atol=1e-5)
