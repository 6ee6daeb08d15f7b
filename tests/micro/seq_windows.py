import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import ba_compare
from oracle import oracle_py as O
from pvio_amd import capi, BAState, BASummary
from pvio_amd.solver import HipContext
O.build(); O.lib()
lib = capi.load(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else capi.load()
def window(seed):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(2, 33))
    kw = dict(n_frames=n, n_landmarks=int(rng.integers(10, 1500)), use_inertial=bool(rng.integers(0, 2)), visibility=int(rng.integers(2, n + 1)),
              plane_fraction=float(rng.choice([0.0, 0.0, 0.3, 0.6])), seed=int(rng.integers(1, 10000)))
    pb = ba_compare.make(O, **kw)
    if rng.random() < 0.4:
        pb.frame_fixed[int(rng.integers(0, n))] = 1
    return pb
def diff(ctx, pb):
    st0, sm0 = BAState(pb), BASummary(pb); O.solve(pb, st0, sm0)
    st1, sm1 = ctx.solve(pb)
    d = np.abs(st1.lm_inv_depth - st0.lm_inv_depth); i = int(np.argmax(d))
    return "worst %.3e at %d (value %.4f)" % (d[i], i, st0.lm_inv_depth[i])
pb37 = window(37)
for first in ([], [36], [35, 36], list(range(30, 37)), list(range(0, 37))):
    ctx = HipContext(lib=lib, device=0)
    for s in first:
        ctx.solve(window(s))
    print("after windows", first[:3], "..." if len(first) > 3 else "", len(first), "->", diff(ctx, pb37), flush=True)
    ctx.close()
