"""Random LARGE windows (too large for the oracle in a fuzz loop) through both landmark roles of k_linearize on the GPU: the register-tile role (linearize_mode 1) and the
large-window role (mode 2, csrc/ba_lin_tp.h) must take the same accept / reject decisions and end within 1e-7 of each other.  Shapes: 3-32 frames, 2000-30 000 landmarks,
random visibility, planes, a fixed frame, landmarks in permuted (unsorted-anchor) order, duplicate blocks."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from pvio_amd import synth
from pvio_amd.solver import HipContext, preintegrate

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
c1, c2 = HipContext(device=0, linearize_mode=1), HipContext(device=0, linearize_mode=2)
bad = 0
for it in range(n_cases):
    n = int(rng.choice([3, 5, 8, 10, 12, 16, 20, 24, 28, 31, 32]))
    vio = bool(rng.integers(0, 2))
    kw = dict(n_frames=n, n_landmarks=int(rng.integers(2000, 30000 if n <= 16 else 9000)), use_inertial=vio, visibility=int(rng.integers(2, n + 1)),
              plane_fraction=float(rng.choice([0.0, 0.0, 0.2])), seed=int(rng.integers(1, 10000)), duplicate_fraction=float(rng.choice([0.0, 0.0, 0.2])))
    pb = synth.make_window(preintegrate=preintegrate if vio else None, **kw)
    if rng.random() < 0.3:
        pb.frame_fixed[int(rng.integers(0, n))] = 1
    if rng.random() < 0.4:
        pb = synth.permute_landmarks(pb, rng.permutation(pb.n_landmarks))
    t0 = time.perf_counter()
    (s1, m1), (s2, m2) = c1.solve(pb), c2.solve(pb)
    t1, t2 = m1.trace(), m2.trace()
    same = len(t1) == len(t2) and all((a["step_is_valid"], a["step_is_successful"]) == (b["step_is_valid"], b["step_is_successful"]) for a, b in zip(t1, t2))
    ds = max(float(np.abs(s1.frame_state - s2.frame_state).max()), float(np.abs(s1.lm_inv_depth - s2.lm_inv_depth).max()))
    dc = abs(m1.final_cost - m2.final_cost) / max(abs(m1.final_cost), 1e-300)
    ok = same and ds <= 1e-7 and dc <= 1e-9 and (s1.lm_valid == s2.lm_valid).all()
    bad += not ok
    print("%2d %-5s %s factors %6d: iterations %d / %d, states differ by %.1e, final cost rel %.1e  %s" % (it, "vio" if vio else "vis", {k: kw[k] for k in ("n_frames", "n_landmarks", "visibility", "plane_fraction", "duplicate_fraction")},
          pb.n_obs, m1.num_iterations, m2.num_iterations, ds, dc, "ok" if ok else "DIFFERENT"), flush=True)
print("FUZZ", "ok" if bad == 0 else "FAILED %d" % bad)
