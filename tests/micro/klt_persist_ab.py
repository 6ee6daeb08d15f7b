"""RECORD OF A DROPPED EXPERIMENT (profiles/r6_klt_persist_ab.txt): the queue-claiming form of k_lk_track_levels and its PVIO_HIP_LK_LEVEL_BLOCKS switch are not in
the tree any more (30 % slower at every setting); with today's library every "levels" variant below is the same kernel.
k_lk_track_levels with 2..6 blocks per CU (fewer blocks than tracks: the rest is claimed from the queue) against k_lk_track and the unit queue on bench.py's
KLT workload (make_image_pair(512, 512, 1500)) and at 3000 / 6000 tracks: mean of 50 launches, interleaved rounds on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, klt_track

variants = {"a wave per track": {"PVIO_HIP_LK_FORM": "1"}, "unit queue": {"PVIO_HIP_LK_FORM": "2"}}
for k in (2, 3, 4, 5, 6, 8):
    variants["levels, %d blocks per CU" % k] = {"PVIO_HIP_LK_FORM": "3", "PVIO_HIP_LK_LEVEL_BLOCKS": str(k)}
ctxs = {}
for name, env in variants.items():
    for key in ("PVIO_HIP_LK_FORM", "PVIO_HIP_LK_LEVEL_BLOCKS"):
        os.environ.pop(key, None)
    os.environ.update(env)
    ctxs[name] = HipContext(device=0)
for n_pts in (1500, 3000, 6000, 600):
    img0, img1, p, truth, init = synth.make_image_pair(512, 512, n_pts)
    imgs = {f: (HipImage(c, img0), HipImage(c, img1)) for f, c in ctxs.items()}
    for f, c in ctxs.items():
        for _ in range(60):
            klt_track(c, imgs[f][0], imgs[f][1], p, init)
    ref = None
    for rnd in range(2):
        for f, c in ctxs.items():
            r = [klt_track(c, imgs[f][0], imgs[f][1], p, init) for _ in range(50)]
            t = np.array([x[2] for x in r])
            ref = ref or r[0]
            same = r[0][0].tobytes() == ref[0].tobytes() and (r[0][1] == ref[1]).all()
            print("%d points, round %d %-26s mean of 50: %.2f us  median %.2f  min %.2f  (%.0f tracks/ms)  %s" % (n_pts, rnd, f, 1e3 * t.mean(), 1e3 * np.median(t), 1e3 * t.min(), n_pts / t.mean(), "bit-identical" if same else "DIFFERENT"), flush=True)
    for f in imgs:
        imgs[f][0].release(), imgs[f][1].release()
