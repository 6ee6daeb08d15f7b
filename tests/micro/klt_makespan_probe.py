"""What a 1500-track LK launch waits for: single-track launch times over the tracks (the chain of one track, no contention), then 1500 copies of the
slowest / of a median track (contention without the spread of the iteration counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, klt_track

form = os.environ.get("PVIO_HIP_LK_FORM", "0")
ctx = HipContext(device=0)
img0, img1, p, truth, init = synth.make_image_pair(512, 512, 1500)
A, B = HipImage(ctx, img0), HipImage(ctx, img1)
def t_of(pp, ii, reps=5):
    return min(klt_track(ctx, A, B, pp, ii)[2] for _ in range(reps)) * 1e3
for _ in range(20):
    klt_track(ctx, A, B, p, init)
single = np.array([t_of(p[k:k + 1], init[k:k + 1], 3) for k in range(1500)])
print("form %s: single-track launches over the 1500 tracks: min %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f us" % (form, single.min(), np.median(single), np.percentile(single, 90), np.percentile(single, 99), single.max()))
print("all 1500 together: %.2f us" % t_of(p, init, 20))
order = np.argsort(single)
for name, k in (("fastest", order[0]), ("median", order[750]), ("p90", order[1350]), ("slowest", order[-1])):
    pp, ii = np.repeat(p[k:k + 1], 1500, 0), np.repeat(init[k:k + 1], 1500, 0)
    print("1500 copies of the %s track (alone %.2f us): %.2f us;  256 copies: %.2f us" % (name, single[k], t_of(pp, ii, 20), t_of(pp[:256], ii[:256], 20)))
# sorted by single-track time, slowest first / last
print("1500 tracks, slowest first: %.2f us   slowest last: %.2f us" % (t_of(p[order[::-1]], init[order[::-1]], 20), t_of(p[order], init[order], 20)))
