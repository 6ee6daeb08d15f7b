"""Shared corner-detection parity check (emulated build and real GPU)."""
import numpy as np

from pvio_amd import synth
from pvio_amd.solver import HipImage, detect_corners


def check_detect(ctx, oracle, width, height, max_corners=1000, quality=1.0e-3, min_distance=20.0):
    img0, _, _, _, _ = synth.make_image_pair(width, height, 8)
    pre = oracle.clahe(img0)  # the detector runs on the preprocessed image (opencv_image.cpp:139 modifies `image` in place)
    r_ref = oracle.harris_response(pre)
    xy_ref, resp_ref = oracle.good_features(r_ref, max_corners, quality, min_distance)
    A = HipImage(ctx, img0, True)
    xy, resp, rmap = detect_corners(ctx, A, max_corners, quality, min_distance, want_response_map=True)
    assert (rmap.view(np.int32) == r_ref.view(np.int32)).all(), "response map differs: max |d| = %g" % np.abs(rmap - r_ref).max()
    assert len(xy) == len(xy_ref) and len(xy) > 0
    assert (xy == xy_ref).all() and (resp.view(np.int32) == resp_ref.view(np.int32)).all()
    A.release()
    return dict(corners=len(xy))
