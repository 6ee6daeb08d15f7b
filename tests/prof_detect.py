"""Where a corner-detection call spends its time (PVIO_KLT_TIMING=1 prints the phases of Klt::detect)."""
import sys, time; sys.path.insert(0, '.')
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, detect_corners
ctx = HipContext(device=0)
for (w, h) in ((512, 512), (752, 480)):
    img0, img1, p, truth, init = synth.make_image_pair(w, h, 100)
    A = HipImage(ctx, img0)
    for _ in range(3):
        detect_corners(ctx, A)
    t0 = time.perf_counter()
    for _ in range(20):
        c, _ = detect_corners(ctx, A)
    print('%dx%d: %.3f ms per call, %d corners' % (w, h, 1e3 * (time.perf_counter() - t0) / 20, len(c)), flush=True)
