"""Ad-hoc corner-detection parity sweep on the GPU: random images (textured, noisy, with flat patches and saturated regions: many
equal responses), random quality / distance / cap -- corners and their order must equal the oracle's exactly."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle_py as O
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, detect_corners
O.build()
ctx = HipContext(device=0)
bad = 0
for seed in range(30):
    rng = np.random.default_rng(7000 + seed)
    w, h = int(rng.choice([200, 320, 512, 752])), int(rng.choice([160, 240, 384, 480]))
    kind = seed % 3
    if kind == 0:
        img = synth.make_image_pair(w, h, 8, seed=int(rng.integers(1, 99999)))[0]
    elif kind == 1:
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
    else:
        img = (rng.integers(0, 4, (h // 8 + 1, w // 8 + 1)) * 85).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w].copy()  # blocks: ties everywhere
    md = float(rng.choice([0.0, 1.0, 3.5, 10.0, 20.0, 41.0])); q = float(rng.choice([1e-3, 1e-2, 0.2])); cap = int(rng.choice([50, 1000]))
    pre = O.clahe(img)
    r_ref = O.harris_response(pre)
    xy_ref, resp_ref = O.good_features(r_ref, cap, q, md)
    A = HipImage(ctx, img, True)
    xy, resp, rmap = detect_corners(ctx, A, cap, q, md, want_response_map=True)
    ok = (rmap.view(np.int32) == r_ref.view(np.int32)).all() and len(xy) == len(xy_ref) and (len(xy) == 0 or ((xy == xy_ref).all() and (resp.view(np.int32) == resp_ref.view(np.int32)).all()))
    bad += 0 if ok else 1
    print(seed, (w, h), 'kind', kind, 'md', md, 'q', q, 'cap', cap, 'corners', len(xy), 'OK' if ok else 'MISMATCH', flush=True)
    A.release()
print('mismatching images:', bad)
