"""Shared KLT parity check (emulated build and real GPU): pyramids bit-exact, LK status identical, positions BIT-IDENTICAL (the kernel sums its
float accumulators in the order oracle_klt.cpp defines: 63 runs of 7 pixels folded by a fixed tree); OpenCV's scalar order within 1e-3 px."""
import numpy as np

from pvio_amd import synth
from pvio_amd.solver import HipImage, klt_track

POS_TOL = 0.0  # px: bit-identical to the oracle in the defined summation order
SCALAR_ORDER_TOL = 1e-3  # px: against OpenCV's scalar left-to-right order (another rounding of the same 441-term sums), SURVEY App. C


def check_klt(ctx, oracle, width, height, n_points, clahe=True):
    img0, img1, p, truth, init = synth.make_image_pair(width, height, n_points)
    c0, c1 = (oracle.clahe(img0), oracle.clahe(img1)) if clahe else (img0, img1)
    P0, P1 = oracle.build_pyramid(c0), oracle.build_pyramid(c1)
    A, B = HipImage(ctx, img0, clahe), HipImage(ctx, img1, clahe)
    for l in range(len(P0)):  # integer pipeline: bit-exact
        gi, gd = A.level(l)
        assert gi.shape == P0[l][0].shape
        assert (gi == P0[l][0]).all(), "level %d image differs" % l
        assert (gd == P0[l][1]).all(), "level %d derivative differs" % l
    n0, s0 = oracle.klt_track(P0, P1, p, init)
    n1, s1, ms = klt_track(ctx, A, B, p, init)
    assert (s0 == s1).all(), "status bytes differ: %d" % int((s0 != s1).sum())
    ok = s0 > 0
    assert np.abs(n0 - n1)[ok].max() <= POS_TOL
    n0s, s0s = oracle.klt_track(P0, P1, p, init, scalar_order=True)
    assert (s0s == s1).all() and np.abs(n0s - n1)[ok].max() <= SCALAR_ORDER_TOL
    err = np.linalg.norm(n1 - truth, axis=1)[ok]
    assert ok.mean() > 0.9 and np.median(err) < 0.25  # sanity: the known homography is recovered (noise floor ~0.05-0.2 px)
    # no initial flow
    n0b, s0b = oracle.klt_track(P0, P1, p, p)
    n1b, s1b, _ = klt_track(ctx, A, B, p, p)
    assert (s0b == s1b).all() and np.abs(n0b - n1b)[s0b > 0].max() <= POS_TOL
    A.release()
    B.release()
    return dict(tracks=int(ok.sum()), max_pos_diff=float(np.abs(n0 - n1)[ok].max()), device_ms=ms)
