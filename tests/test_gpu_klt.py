"""GPU parity tests of the KLT front end (C-ABI of libpvio_hip.so) against the CPU oracle.

Bar: CLAHE / pyramid / Scharr levels bit-exact (integer + strictly ordered float32); LK status bytes identical and positions
BIT-IDENTICAL: the kernel reduces its float32 accumulators in the order oracle/oracle_klt.cpp defines (63 runs of 7 pixels, a fixed fold
tree), so nothing is left to "rounding" between the two -- OpenCV's own scalar order is a third rounding of the same sums (<= 1e-3 px)."""
import numpy as np
import pytest

import klt_compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ctx():
    from pvio_amd.solver import HipContext
    ctx = HipContext(device=0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("size", [(752, 480, 1500), (512, 512, 1500), (320, 240, 300), (175, 131, 40)])
def test_gpu_klt_matches_oracle(gpu_ctx, oracle, size):
    print(size, klt_compare.check_klt(gpu_ctx, oracle, *size))


def test_gpu_klt_large_batch(gpu_ctx, oracle):
    print(klt_compare.check_klt(gpu_ctx, oracle, 512, 512, 3000))


def test_gpu_klt_no_clahe_and_empty(gpu_ctx, oracle):
    import numpy as np
    from pvio_amd import synth
    from pvio_amd.solver import HipImage, klt_track
    klt_compare.check_klt(gpu_ctx, oracle, 320, 240, 100, clahe=False)
    img0, img1, p, truth, init = synth.make_image_pair(320, 240, 10)
    A, B = HipImage(gpu_ctx, img0), HipImage(gpu_ctx, img1)
    q, st, _ = klt_track(gpu_ctx, A, B, p[:0], init[:0])  # empty input
    assert q.shape == (0, 2) and st.shape == (0,)
    # identical images: zero motion
    q, st, _ = klt_track(gpu_ctx, A, A, p, p)
    assert st.all() and np.abs(q - p).max() < 1e-3  # (LK does not return the start exactly: its first step is a few 1e-5 px)


def test_gpu_corner_detection_matches_oracle(gpu_ctx, oracle):
    import gftt_compare
    print(gftt_compare.check_detect(gpu_ctx, oracle, 752, 480))   # EuRoC size
    print(gftt_compare.check_detect(gpu_ctx, oracle, 512, 512, max_corners=300, min_distance=11.0))  # TUM-VI size
    print(gftt_compare.check_detect(gpu_ctx, oracle, 333, 241, quality=0.05))


@pytest.mark.parametrize("md,cap", [(7.5, 1000), (1.0, 1000), (0.0, 300), (20.0, 60), (45.0, 1000)])
def test_gpu_corner_distance_filter_is_exact(gpu_ctx, oracle, md, cap):
    import gftt_compare
    print(md, cap, gftt_compare.check_detect(gpu_ctx, oracle, 512, 384, max_corners=cap, min_distance=md))


# ---- bounded versions of the ad-hoc sweeps (tests/sweep_random_klt.py, tests/sweep_random_detect.py) ---------------------------
@pytest.mark.gpu
def test_gpu_lk_random_pairs_status_exact_positions_bounded(oracle):
    """Eight random pairs (sizes, motion up to 25 px, noise, a flat patch in every third pair, 300 extra points anywhere with poor
    initial guesses; ~8 800 points): status bytes AND positions are identical to the oracle's, bit for bit -- the nearly flat regions included,
    where under another summation order the iteration takes another exit (the scalar-order oracle shows that: a handful of points, 0.1 px)."""
    from pvio_amd import synth
    from pvio_amd.solver import HipContext, HipImage, klt_track
    ctx = HipContext(device=0)
    tot = outl = scalar_flips = scalar_outl = 0
    worst = scalar_worst = 0.0
    for seed in (0, 1, 2, 3, 6, 9, 12, 15):
        rng = np.random.default_rng(9000 + seed)
        w, h = int(rng.choice([320, 512, 640, 752])), int(rng.choice([240, 384, 480, 512]))
        try:
            img0, img1, p, truth, init = synth.make_image_pair(w, h, 800, seed=int(rng.integers(1, 100000)), max_motion=float(rng.choice([6.0, 12.0, 25.0])),
                                                               noise_sigma=float(rng.choice([0.0, 2.0, 8.0])))
        except AssertionError:
            continue
        extra = np.column_stack([rng.uniform(0, w, 300), rng.uniform(0, h, 300)]).astype(np.float32)
        p2 = np.vstack([p, extra]).astype(np.float32)
        init2 = np.vstack([init, extra + rng.uniform(-8, 8, extra.shape).astype(np.float32)]).astype(np.float32)
        if seed % 3 == 0:
            img0, img1 = img0.copy(), img1.copy()
            img0[h // 4:h // 2, w // 4:w // 2] = 128
            img1[h // 4:h // 2, w // 4:w // 2] = 128
        clahe = bool(seed % 2)
        c0, c1 = (oracle.clahe(img0), oracle.clahe(img1)) if clahe else (img0, img1)
        P0, P1 = oracle.build_pyramid(c0), oracle.build_pyramid(c1)
        A, B = HipImage(ctx, img0, clahe), HipImage(ctx, img1, clahe)
        n0, s0 = oracle.klt_track(P0, P1, p2, init2)
        n1, s1, _ = klt_track(ctx, A, B, p2, init2)
        A.release(), B.release()
        assert (s0 == s1).all(), (seed, int((s0 != s1).sum()))
        ok = s0 > 0
        d = np.abs(n0 - n1)[ok].max(axis=1) if ok.any() else np.zeros(0)
        tot += int(ok.sum())
        outl += int((d > 0).sum())
        worst = max(worst, float(d.max(initial=0.0)))
        n0s, s0s = oracle.klt_track(P0, P1, p2, init2, scalar_order=True)  # OpenCV's scalar order: same decisions, other rounding
        both = ok & (s0s > 0)
        ds = np.abs(n0s - n1)[both].max(axis=1) if both.any() else np.zeros(0)
        scalar_flips += int((s0s != s1).sum())
        scalar_outl += int((ds > 1e-3).sum())
        scalar_worst = max(scalar_worst, float(ds.max(initial=0.0)))
    ctx.close()
    print("LK sweep: %d tracked points, %d not bit-identical (worst %.3g px); against the scalar order: %d status flips, %d beyond 1e-3 px, worst %.3g px" % (
        tot, outl, worst, scalar_flips, scalar_outl, scalar_worst))
    assert tot > 4000 and outl == 0 and worst == 0.0
    assert scalar_flips <= 0.002 * tot and scalar_outl <= 0.002 * tot and scalar_worst < 0.15


@pytest.mark.gpu
def test_gpu_corner_detection_random_images_exact(oracle):
    """Nine random images (texture, uniform noise, tie-heavy block patterns; random quality level / minimum distance / cap):
    the Harris response map is bit-identical and the selected corners, their order and responses are exactly the oracle's."""
    from pvio_amd import synth
    from pvio_amd.solver import HipContext, HipImage, detect_corners
    ctx = HipContext(device=0)
    for seed in range(0, 27, 3):
        rng = np.random.default_rng(7000 + seed)
        w, h = int(rng.choice([200, 320, 512, 752])), int(rng.choice([160, 240, 384, 480]))
        kind = (seed // 3) % 3
        if kind == 0:
            img = synth.make_image_pair(w, h, 8, seed=int(rng.integers(1, 99999)))[0]
        elif kind == 1:
            img = rng.integers(0, 256, (h, w)).astype(np.uint8)
        else:
            img = (rng.integers(0, 4, (h // 8 + 1, w // 8 + 1)) * 85).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w].copy()
        md, q, cap = float(rng.choice([0.0, 1.0, 3.5, 10.0, 20.0, 41.0])), float(rng.choice([1e-3, 1e-2, 0.2])), int(rng.choice([50, 1000]))
        r_ref = oracle.harris_response(oracle.clahe(img))
        xy_ref, resp_ref = oracle.good_features(r_ref, cap, q, md)
        A = HipImage(ctx, img, True)
        xy, resp, rmap = detect_corners(ctx, A, cap, q, md, want_response_map=True)
        A.release()
        assert (rmap.view(np.int32) == r_ref.view(np.int32)).all(), seed
        assert len(xy) == len(xy_ref) and (len(xy) == 0 or ((xy == xy_ref).all() and (resp.view(np.int32) == resp_ref.view(np.int32)).all())), seed
    ctx.close()


def test_gpu_lk_border_corner_starts_are_bit_identical(gpu_ctx, oracle):
    """Round 5, from the diagnosis of the long-sequence run (test_dropin_sequence.py::test_long_sequence_*): corners detected exactly ON the 20-pixel
    border -- (20, 294) in frame 62 of that sequence -- are legal keypoints whose windows reach into the padding at every coarse level, and whose LK result
    is killed by the output gate as soon as it moves outward.  Frames 62 -> 63 of that sequence: a dense grid of 243 starts around that corner and 207
    points along the whole border ring, no initial flow: status bytes identical, positions bit-identical between the oracle and the kernel.  (What did
    separate the two runs of that sequence was a tie in the F-matrix RANSAC, not LK: every LK call's inputs and outputs were bit-identical.)"""
    import test_host_headless as hh
    from pvio_amd.solver import HipImage, klt_track
    images, *_ = hh.render_sequence(64, relief=True, sweep=True)
    i0, i1 = images[62], images[63]
    P0, P1 = oracle.build_pyramid(oracle.clahe(i0)), oracle.build_pyramid(oracle.clahe(i1))
    A, B = HipImage(gpu_ctx, i0, True), HipImage(gpu_ctx, i1, True)
    offs = np.array([(dx, dy) for dx in np.linspace(-4, 4, 9) for dy in np.linspace(-4, 4, 9)], np.float32)
    corner = np.array([20.0, 294.0], np.float32)
    n_alive = 0
    for target in (corner, np.array([21.228, 294.551], np.float32), np.array([75.523, 283.173], np.float32)):
        prev = np.repeat(corner[None, :], len(offs), 0)
        init = (target[None, :] + offs).astype(np.float32)
        n0, s0 = oracle.klt_track(P0, P1, prev, init)
        n1, s1, _ = klt_track(gpu_ctx, A, B, prev, init)
        assert (s0 == s1).all() and (n0[s0 > 0] == n1[s0 > 0]).all()
        n_alive += int((s0 > 0).sum())
    # points along the whole 20-pixel border and just inside it, no initial flow
    xs, ys = np.arange(20, hh.W - 20, 12, dtype=np.float32), np.arange(20, hh.H - 20, 12, dtype=np.float32)
    ring = np.array([(x, y) for x in xs for y in (20.0, 21.5, hh.H - 21.0)] + [(x, y) for y in ys for x in (20.0, 21.5, hh.W - 21.0)], np.float32)
    n0, s0 = oracle.klt_track(P0, P1, ring, ring)
    n1, s1, _ = klt_track(gpu_ctx, A, B, ring, ring)
    assert (s0 == s1).all() and (n0[s0 > 0] == n1[s0 > 0]).all()
    print("border corner: 243 starts (%d alive) + %d border points (%d alive): status identical, positions bit-identical" % (n_alive, len(ring), int((s0 > 0).sum())))
    A.release()
    B.release()


def test_gpu_lk_forms_are_bit_identical(oracle, monkeypatch):
    """The three launch forms of the LK search, each forced by PVIO_HIP_LK_FORM in a context of its own -- k_lk_track_levels (a workgroup per track,
    a wave per pyramid level, hardware barriers between the levels' searches: the default since round 6), k_lk_track (a wave per track),
    k_lk_track_units ((track, level) units from a queue in LDS, eight waves per CU) -- and the oracle: status bytes identical, positions
    bit-identical, for track counts from fewer than CUs to six per SIMD.  The first build of the unit kernel hung the GPU
    (profiles/r5_ab_klt_units_hang.txt: the compiler threaded the two `lane == 0` branches across the loop's back edge); the per-test time limit
    bounds this test should that ever come back."""
    from pvio_amd import synth
    from pvio_amd.solver import HipContext, HipImage, klt_track
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "1")
    per_track = HipContext(device=0)
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "2")
    units = HipContext(device=0)
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "3")
    levels = HipContext(device=0)
    img0, img1, p, truth, init = synth.make_image_pair(512, 512, 6000)
    P0, P1 = oracle.build_pyramid(oracle.clahe(img0)), oracle.build_pyramid(oracle.clahe(img1))
    imgs = [(HipImage(c, img0), HipImage(c, img1)) for c in (per_track, units, levels)]
    for n in (64, 5, 1500, 1025, 6000, 257):
        qa, sa, _ = klt_track(per_track, imgs[0][0], imgs[0][1], p[:n], init[:n])
        qb, sb, _ = klt_track(units, imgs[1][0], imgs[1][1], p[:n], init[:n])
        qc, sc, _ = klt_track(levels, imgs[2][0], imgs[2][1], p[:n], init[:n])
        assert (sa == sb).all() and qa.tobytes() == qb.tobytes(), n
        assert (sa == sc).all() and qa.tobytes() == qc.tobytes(), n
        if n <= 1500:
            q0, s0 = oracle.klt_track(P0, P1, p[:n], init[:n])
            assert (s0 == sc).all() and np.abs(q0 - qc)[s0 > 0].max() == 0.0
    for c in (per_track, units, levels):
        c.close()
