"""The LK launch forms on EXACTLY bench.py's KLT workload (make_image_pair(512, 512, 1500): its own point set, not the first 1500 of a 6000-point grid),
mean of 50 launches, interleaved rounds on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, klt_track

NAMES = {"3": "levels (WG per track)", "1": "a wave per track", "2": "unit queue", "0": "default"}
ctxs = {}
for f in ("1", "2", "3", "0"):
    os.environ["PVIO_HIP_LK_FORM"] = f
    ctxs[f] = HipContext(device=0)
SIZES = [int(a) for a in sys.argv[1:]] or [1500, 150, 300]  # (other sizes: the crossover of the launch rule)
for n_pts in SIZES:
    img0, img1, p, truth, init = synth.make_image_pair(512, 512, n_pts)
    imgs = {f: (HipImage(c, img0), HipImage(c, img1)) for f, c in ctxs.items()}
    for f, c in ctxs.items():
        for _ in range(100):
            klt_track(c, imgs[f][0], imgs[f][1], p, init)
    ref = None
    for rnd in range(3):
        for f, c in ctxs.items():
            r = [klt_track(c, imgs[f][0], imgs[f][1], p, init) for _ in range(50)]
            t = np.array([x[2] for x in r])
            ref = ref or r[0]
            same = r[0][0].tobytes() == ref[0].tobytes() and (r[0][1] == ref[1]).all()
            print("%d points, round %d %-22s mean of 50: %.2f us  median %.2f  min %.2f  (%.0f tracks/ms)  %s" % (n_pts, rnd, NAMES[f], 1e3 * t.mean(), 1e3 * np.median(t), 1e3 * t.min(), n_pts / t.mean(), "bit-identical" if same else "DIFFERENT"), flush=True)
