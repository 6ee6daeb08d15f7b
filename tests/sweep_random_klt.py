"""Ad-hoc LK parity sweep on the GPU: image pairs of random seed / size / motion / noise, points also near borders and in flat
regions (status decisions at the thresholds): status bytes must equal the oracle's, positions within 1e-3 px."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle_py as O
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, klt_track
O.build()
ctx = HipContext(device=0)
tot = mism = 0
worst = 0.0
for seed in range(24):
    rng = np.random.default_rng(9000 + seed)
    w, h = int(rng.choice([320, 512, 640, 752])), int(rng.choice([240, 384, 480, 512]))
    try:
        img0, img1, p, truth, init = synth.make_image_pair(w, h, 800, seed=int(rng.integers(1, 100000)), max_motion=float(rng.choice([6.0, 12.0, 25.0])),
                                                           noise_sigma=float(rng.choice([0.0, 2.0, 8.0])))
    except AssertionError:
        continue  # the generator refuses homographies that move a point further than asked
    # extra points anywhere in the image (borders, wherever) with poor initial guesses
    extra = np.column_stack([rng.uniform(0, w, 300), rng.uniform(0, h, 300)]).astype(np.float32)
    p2 = np.vstack([p, extra]).astype(np.float32)
    init2 = np.vstack([init, extra + rng.uniform(-8, 8, extra.shape).astype(np.float32)]).astype(np.float32)
    if seed % 3 == 0:  # a flat patch: degenerate gradient matrices
        img0 = img0.copy(); img1 = img1.copy()
        img0[h // 4:h // 2, w // 4:w // 2] = 128; img1[h // 4:h // 2, w // 4:w // 2] = 128
    clahe = bool(seed % 2)
    c0, c1 = (O.clahe(img0), O.clahe(img1)) if clahe else (img0, img1)
    P0, P1 = O.build_pyramid(c0), O.build_pyramid(c1)
    A, B = HipImage(ctx, img0, clahe), HipImage(ctx, img1, clahe)
    n0, s0 = O.klt_track(P0, P1, p2, init2)
    n1, s1, _ = klt_track(ctx, A, B, p2, init2)
    bad = int((s0 != s1).sum())
    ok = (s0 > 0) & (s1 > 0)
    d = float(np.abs(n0 - n1)[ok].max()) if ok.any() else 0.0
    tot += len(s0); mism += bad; worst = max(worst, d)
    print(seed, (w, h), 'tracks', int(ok.sum()), 'of', len(s0), 'status mismatches', bad, 'max pos diff %.2e' % d, flush=True)
    A.release(); B.release()
print('total points', tot, 'status mismatches', mism, 'worst position difference %.2e' % worst)
