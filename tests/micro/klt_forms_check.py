"""The three launch forms of the LK search on the GPU in one process (PVIO_HIP_LK_FORM: 0 k_lk_track_levels, 1 k_lk_track, 2 k_lk_track_units):
bit-identical positions / status bytes, and the launch times as bench.py reports them (MEAN of 50 hipEvent pairs) next to min / median.
Run under a hard time limit (a hung kernel does not end by itself): timeout -s KILL 60 python tests/micro/klt_forms_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, klt_track

NAMES = {3: "levels (WG per track)", 1: "a wave per track", 2: "unit queue"}
ctxs = {}
for f in (1, 2, 3):
    os.environ["PVIO_HIP_LK_FORM"] = str(f)
    ctxs[f] = HipContext(device=0)
img0, img1, p, truth, init = synth.make_image_pair(512, 512, 6000)
imgs = {f: (HipImage(c, img0), HipImage(c, img1)) for f, c in ctxs.items()}
print("contexts up", flush=True)
bad = 0
for n in (64, 5, 1500, 1025, 3000, 6000, 257, 150, 1500):
    res = {}
    for f in (1, 2, 3):
        res[f] = klt_track(ctxs[f], imgs[f][0], imgs[f][1], p[:n], init[:n])
        print("n %5d %-22s %.1f us" % (n, NAMES[f], 1e3 * res[f][2]), flush=True)
    same = all(res[f][0].tobytes() == res[1][0].tobytes() and (res[f][1] == res[1][1]).all() for f in (3, 2))
    bad += not same
    print("   ", "bit-identical" if same else "DIFFERENT", flush=True)
for n in (150, 512, 1024, 1500, 2048, 3000, 6000):
    for rnd in range(2):  # interleaved rounds: box drift shows as a difference between the two rounds of one form
        for f in (1, 2, 3):
            t = np.array([klt_track(ctxs[f], imgs[f][0], imgs[f][1], p[:n], init[:n])[2] for _ in range(50)])
            print("n %5d round %d %-22s mean of 50: %.2f us  median %.2f  min %.2f   (%.0f tracks/ms by the mean)" % (n, rnd, NAMES[f], 1e3 * t.mean(), 1e3 * np.median(t), 1e3 * t.min(), n / t.mean()), flush=True)
print("CHECK", "ok" if bad == 0 else "FAILED")
