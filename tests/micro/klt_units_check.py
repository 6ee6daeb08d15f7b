"""k_lk_track_units against k_lk_track on the GPU in one process: bit-identical positions / status bytes and the launch times.
Run under a hard time limit (a hung kernel does not end by itself): timeout -s KILL 20 python tests/micro/klt_units_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, klt_track

os.environ["PVIO_HIP_LK_UNITS"] = "0"
per_track = HipContext(device=0)
os.environ["PVIO_HIP_LK_UNITS"] = "1"
units = HipContext(device=0)
img0, img1, p, truth, init = synth.make_image_pair(512, 512, 6000)
A0, B0 = HipImage(per_track, img0), HipImage(per_track, img1)
A1, B1 = HipImage(units, img0), HipImage(units, img1)
print("contexts up", flush=True)
bad = 0
for n in (64, 5, 1500, 1025, 3000, 6000, 257, 1500):
    qa, sa, ta = klt_track(per_track, A0, B0, p[:n], init[:n])
    print("n %5d a wave per track %.1f us ..." % (n, 1e3 * ta), end="", flush=True)
    qb, sb, tb = klt_track(units, A1, B1, p[:n], init[:n])
    same = qa.tobytes() == qb.tobytes() and (sa == sb).all()
    bad += not same
    print(" units %.1f us  %s" % (1e3 * tb, "bit-identical" if same else "DIFFERENT"), flush=True)
for n in (1500, 2048, 3000, 6000):
    ta = min(klt_track(per_track, A0, B0, p[:n], init[:n])[2] for _ in range(8))
    tb = min(klt_track(units, A1, B1, p[:n], init[:n])[2] for _ in range(8))
    print("n %5d  min of 8: a wave per track %.1f us   units %.1f us   (%.0f -> %.0f tracks/ms)" % (n, 1e3 * ta, 1e3 * tb, n / ta, n / tb), flush=True)
print("CHECK", "ok" if bad == 0 else "FAILED")
