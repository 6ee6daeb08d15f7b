"""k_lk_track_levels variants against k_lk_track on one box: mean of 50 launches (what bench.py reports), interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, klt_track

variants = {"wave per track": {"PVIO_HIP_LK_FORM": "1"}, "levels": {"PVIO_HIP_LK_FORM": "3"}, "default": {"PVIO_HIP_LK_FORM": "0"}}
ctxs = {}
for name, env in variants.items():
    for k in ("PVIO_HIP_LK_FORM", "PVIO_HIP_LK_PRIO"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctxs[name] = HipContext(device=0)
img0, img1, p, truth, init = synth.make_image_pair(512, 512, 6000)
imgs = {k: (HipImage(c, img0), HipImage(c, img1)) for k, c in ctxs.items()}
ref = None
for n in (150, 1024, 1500, 2048, 3000, 6000):
    for rnd in range(2):
        for name, c in ctxs.items():
            r = [klt_track(c, imgs[name][0], imgs[name][1], p[:n], init[:n]) for _ in range(50)]
            t = np.array([x[2] for x in r])
            if name == "wave per track":
                ref = r[0]
            same = r[0][0].tobytes() == ref[0].tobytes() and (r[0][1] == ref[1]).all()
            print("n %5d round %d %-22s mean of 50: %.2f us  median %.2f  min %.2f  (%.0f tracks/ms)  %s" % (n, rnd, name, 1e3 * t.mean(), 1e3 * np.median(t), 1e3 * t.min(), n / t.mean(), "bit-identical" if same else "DIFFERENT"), flush=True)
