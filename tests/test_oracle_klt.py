"""Invariants that pin the KLT oracle (no OpenCV available here -> parity unpinned against the real reference)."""
import numpy as np

from pvio_amd import synth


def test_pyr_down_and_scharr_known_answers(oracle):
    # constant image: pyrDown keeps the constant ((16*16*c + 128) >> 8 == c), Scharr is zero
    img = np.full((40, 56), 77, np.uint8)
    lv = oracle.build_pyramid(np.pad(img, 0))
    assert all((l[0] == 77).all() and (l[1] == 0).all() for l in lv)
    # horizontal ramp: dI/dx = 32 * slope (Scharr gain 2 * 16), dI/dy = 0 in the interior
    ramp = np.tile((np.arange(64) * 2).astype(np.uint8), (48, 1))
    d = oracle.build_pyramid(ramp)[0][1]
    assert (d[2:-2, 2:-2, 0] == 2 * 32).all() and (d[2:-2, 2:-2, 1] == 0).all()
    # reflect-101 border: derivative across the border is zero for the mirrored direction
    assert (d[:, 0, 0] == 0).all() and (d[:, -1, 0] == 0).all()
    # impulse: pyrDown is the separable [1 4 6 4 1]/16 kernel
    imp = np.zeros((44, 44), np.uint8)
    imp[20, 20] = 255
    p1 = oracle.build_pyramid(imp)[1][0]
    assert p1[10, 10] == (36 * 255 + 128) >> 8 and p1[10, 9] == (6 * 255 + 128) >> 8 and p1[9, 9] == (1 * 255 + 128) >> 8


def test_clahe_properties(oracle):
    rng = np.random.default_rng(0)
    img = (rng.normal(120, 10, size=(96, 128))).clip(0, 255).astype(np.uint8)
    out = oracle.clahe(img)
    assert out.std() > 2.0 * img.std()          # contrast is stretched
    # monotone within a tile centre: a brighter input pixel never maps to a darker output at the same location
    flat = np.full((64, 64), 100, np.uint8)
    o2 = oracle.clahe(flat)
    assert o2.min() == o2.max()                 # a flat image stays flat
    # sizes that are not multiples of the 8x8 tile grid are padded (BORDER_REFLECT_101), not rejected
    odd = (rng.uniform(0, 255, size=(75, 101))).astype(np.uint8)
    assert oracle.clahe(odd).shape == odd.shape


def test_lk_recovers_known_homography(oracle):
    img0, img1, p, truth, init = synth.make_image_pair(320, 240, 200)
    P0, P1 = oracle.build_pyramid(oracle.clahe(img0)), oracle.build_pyramid(oracle.clahe(img1))
    nxt, st = oracle.klt_track(P0, P1, p, init)
    assert st.mean() > 0.95
    err = np.linalg.norm(nxt - truth, axis=1)[st > 0]
    assert np.median(err) < 0.1 and err.max() < 1.0
    # no initial flow given (next == prev) still converges for <= 6 px motion
    nxt2, st2 = oracle.klt_track(P0, P1, p, p)
    err2 = np.linalg.norm(nxt2 - truth, axis=1)[st2 > 0]
    assert st2.mean() > 0.9 and np.median(err2) < 0.1
    # identical images: zero motion, every track survives
    nxt3, st3 = oracle.klt_track(P0, P0, p, p)
    assert st3.all() and np.abs(nxt3 - p).max() < 1e-3


def test_lk_status_rules(oracle):
    img0, img1, p, truth, init = synth.make_image_pair(320, 240, 50)
    P0, P1 = oracle.build_pyramid(img0), oracle.build_pyramid(img1)
    # a point tracked into the 20-px border band is dropped (opencv_image.cpp:104-109)
    pts = np.array([[25.0, 120.0], [160.0, 120.0]], np.float32)
    guess = np.array([[10.0, 120.0], [160.0, 120.0]], np.float32)
    nxt, st = oracle.klt_track(P0, P0, pts, guess)
    assert st[1] == 1
    # textureless image: minimum-eigenvalue test fails at level 0 -> status 0
    flat = np.full((240, 320), 128, np.uint8)
    F = oracle.build_pyramid(flat)
    _, stf = oracle.klt_track(F, F, pts[1:], pts[1:])
    assert stf[0] == 0
