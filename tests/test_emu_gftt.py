"""Corner detection (Harris / goodFeaturesToTrack) through the kernel emulator; the GPU versions are in
tests/test_gpu_klt.py.  The oracle (oracle/oracle_gftt.cpp) fixes one order of the float operations: the device response
map has to equal it bit for bit, the selected corners exactly."""
import os
import subprocess

import numpy as np
import pytest

import gftt_compare
from pvio_amd import capi
from pvio_amd.solver import HipContext

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu")


@pytest.fixture(scope="module")
def emu_ctx():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    ctx = HipContext(lib=capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so")))
    yield ctx
    ctx.close()


def test_oracle_harris_known_answers(oracle):
    # a constant image and a pure ramp have no corners; an isolated bright square has exactly four
    assert np.all(oracle.harris_response(np.full((48, 64), 77, np.uint8)) == 0)
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 3, (48, 1))
    r = oracle.harris_response(ramp)
    assert r.max() <= 0  # edges only: a c - b^2 = 0, the trace term makes the response negative
    xy, _ = oracle.good_features(r, 100, 1e-3, 5.0)
    assert len(xy) == 0
    sq = np.zeros((64, 64), np.uint8)
    sq[20:44, 20:44] = 200
    r = oracle.harris_response(sq)
    xy, resp = oracle.good_features(r, 100, 0.1, 5.0)
    assert len(xy) == 4 and np.all(np.diff(resp) <= 0)
    for cx, cy in ((20, 20), (43, 20), (20, 43), (43, 43)):
        assert np.min(np.abs(xy - [cx, cy]).sum(1)) <= 2
    # minimum distance: no two selected corners closer than asked
    rng = np.random.default_rng(3)
    img = (rng.uniform(0, 255, (96, 128))).astype(np.uint8)
    xy, _ = oracle.good_features(oracle.harris_response(img), 1000, 1e-3, 9.0)
    d = np.linalg.norm(xy[:, None] - xy[None], axis=2) + np.eye(len(xy)) * 1e9
    assert len(xy) > 20 and d.min() >= 9.0


def test_emulated_detection_matches_oracle(emu_ctx, oracle):
    gftt_compare.check_detect(emu_ctx, oracle, 160, 120)


def test_emulated_detection_odd_size_small_distance(emu_ctx, oracle):
    gftt_compare.check_detect(emu_ctx, oracle, 151, 117, max_corners=50, min_distance=7.0)


@pytest.mark.parametrize("md,cap", [(7.5, 1000), (1.0, 1000), (0.0, 200), (33.0, 1000)])
def test_emulated_detection_distance_filter_is_exact(emu_ctx, oracle, md, cap):
    """The device drops candidates that the greedy minimum-distance walk is certain to reject (dominated ones); whatever the
    distance -- fractional, 1, none (no filter), larger than the tile -- the corners must be the oracle's, in its order."""
    gftt_compare.check_detect(emu_ctx, oracle, 128, 96, max_corners=cap, min_distance=md)
