"""LK launch time against the number of tracks (latency- or throughput-bound?)."""
import sys; sys.path.insert(0, '.')
import numpy as np
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, klt_track
ctx = HipContext(device=0)
img0, img1, p, truth, init = synth.make_image_pair(512, 512, 6000)
A, B = HipImage(ctx, img0), HipImage(ctx, img1)
for n in (64, 256, 512, 1024, 1500, 2048, 3000, 6000):
    ms = []
    for _ in range(6):
        q, st, t = klt_track(ctx, A, B, p[:n], init[:n])
        ms.append(t)
    print(n, 'tracks  min %.1f us  median %.1f us  ok %.3f' % (1e3 * min(ms), 1e3 * float(np.median(ms)), st.mean()), flush=True)
# exact initial guess: how much is iteration count?
for n in (1500,):
    ms = [klt_track(ctx, A, B, p[:n], truth[:n].astype(np.float32))[2] for _ in range(6)]
    print(n, 'tracks, initial guess = truth: min %.1f us' % (1e3 * min(ms)))
