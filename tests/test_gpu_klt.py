"""GPU parity tests of the KLT front end (C-ABI of libpvio_hip.so) against the CPU oracle.

Bar: CLAHE / pyramid / Scharr levels bit-exact (integer + strictly ordered float32); LK status bytes identical and
positions within 1e-3 px (float32 accumulators are reduced as a wave butterfly instead of left-to-right)."""
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
    assert st.all() and np.abs(q - p).max() < 1e-3


def test_gpu_corner_detection_matches_oracle(gpu_ctx, oracle):
    import gftt_compare
    print(gftt_compare.check_detect(gpu_ctx, oracle, 752, 480))   # EuRoC size
    print(gftt_compare.check_detect(gpu_ctx, oracle, 512, 512, max_corners=300, min_distance=11.0))  # TUM-VI size
    print(gftt_compare.check_detect(gpu_ctx, oracle, 333, 241, quality=0.05))


@pytest.mark.parametrize("md,cap", [(7.5, 1000), (1.0, 1000), (0.0, 300), (20.0, 60), (45.0, 1000)])
def test_gpu_corner_distance_filter_is_exact(gpu_ctx, oracle, md, cap):
    import gftt_compare
    print(md, cap, gftt_compare.check_detect(gpu_ctx, oracle, 512, 384, max_corners=cap, min_distance=md))
