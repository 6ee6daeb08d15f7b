"""Ad-hoc sweep of the drop-in A/B (tests/test_dropin_map.py): windows of random shape built as the reference's REAL pvio::Map and solved by the reference's own
BundleAdjustor (libpvio_ref.so) and by the product's adapter linked in its place (libpvio_dropin*.so), every Frame / Track / flag / Plane::tracks compared.
usage: python tests/sweep_random_dropin.py [emu|gpu] [count]      (emu: kernels in the fiber emulator, CPU only)
The reference side solves with mini-Ceres' DENSE Cholesky of every unknown, so landmarks are kept <= 400 (seconds per window)."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
import ba_compare
import test_dropin_map as T
from oracle import oracle_py as O
from oracle import ref_py

kind = sys.argv[1] if len(sys.argv) > 1 else "emu"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
O.build()
ref_py.lib()
drop = ref_py.dropin(kind)
bad = 0
worst = 0.0
for seed in range(count):
    rng = np.random.default_rng(8000 + seed)
    n = int(rng.integers(3, 13))
    kw = dict(n_frames=n, n_landmarks=int(rng.integers(20, 400)), use_inertial=bool(rng.integers(0, 2)), visibility=int(rng.integers(2, n + 1)),
              plane_fraction=float(rng.choice([0.0, 0.0, 0.3, 0.5])), seed=int(rng.integers(1, 10000)))
    if rng.random() < 0.25:
        kw["duplicate_fraction"] = 0.3
        kw["plane_fraction"] = 0.0
    pb = ba_compare.make(O, **kw)
    try:
        r = T.diff_solve(ref_py, drop, pb)
        worst = max(worst, r["worst_state_diff"], r["worst_inv_depth"])
        print(seed, kw, 'ok', '%.1e' % r['worst_state_diff'], '%.1e' % r['worst_inv_depth'], r['iterations'], flush=True)
    except AssertionError as e:
        bad += 1
        print(seed, kw, 'FAIL', str(e)[:300], flush=True)
print('failures:', bad, 'of', count, 'worst difference', '%.2e' % worst)
