"""Ad-hoc F-matrix RANSAC sweep on the GPU (round 5): random two-view scenes of random size, outlier share and noise -- plus the correspondences of the two
sequence frames at which hypotheses tied (tests/golden/ransac_ties.npz) -- through pvio_hip_fundamental_ransac (hypotheses on the device) and through the
oracle's entry point in the DEFINED arithmetic (oracle_ransac.cpp, namespace defined): inlier count, mask and every bit of the winning matrix must be equal;
against the oracle's independent entry point (Jacobi null space, libm closed form) the masks may differ at threshold ties: counted and reported."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import os
import numpy as np
import test_host_ransac as T
from oracle import oracle_py as O
from pvio_amd.solver import HipContext, fundamental_ransac
O.build()
ctx = HipContext(device=0)
cases = []
rng = np.random.default_rng(4242)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    n = int(rng.choice([8, 20, 60, 130, 300, 700, 1500]))
    p, q, _ = T.two_views(n, float(rng.choice([0.0, 0.1, 0.3, 0.5])), float(rng.choice([0.05, 0.3, 0.8, 1.5])), int(rng.integers(1, 1 << 30)))
    cases.append((p, q))
z = np.load(os.path.join("tests", "golden", "ransac_ties.npz"))
cases += [(z["p_a"], z["q_a"]), (z["p_b"], z["q_b"])]
bad = ties = hyp_total = 0
for i, (p, q) in enumerate(cases):
    good, mask, F, hyp = fundamental_ransac(ctx, p, q)
    d_good, d_mask, d_F = T.run_oracle_defined(p, q)
    o_good, o_mask, _ = T.run_oracle(p, q)
    hyp_total += hyp
    same = good == d_good and (mask.astype(bool) == d_mask).all() and (good == 0 or (F == d_F).all())
    if not same:
        bad += 1
        print("case", i, "n", len(p), "DIFFERS from the defined arithmetic:", good, d_good, int((mask.astype(bool) != d_mask).sum()))
    if (mask.astype(bool) != o_mask).any():
        ties += 1
print("%d cases, %d hypotheses on the device: %d differ from the oracle in the defined arithmetic (count / mask / matrix bits); %d differ from the independent "
      "oracle (threshold ties between null-space algorithms)" % (len(cases), hyp_total, bad, ties))
sys.exit(1 if bad else 0)
