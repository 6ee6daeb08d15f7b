"""world_size-2 (3, 4, 8) CPU test of the landmark-sharded bundle adjustment: gloo stands in for RCCL, the fiber emulator
for the GPU.  Every rank must take the same decisions and reproduce the unsharded oracle."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import ba_compare
from pvio_amd import BAState, BASummary

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# mode 2 = the matrix-core form of k_linearize (what sharded large windows run: bench.py's scaling_window leg)
@pytest.mark.parametrize("case,world,mode", [("vio_plane", 2, 0), ("vision_partial", 3, 0), ("vio_partial", 2, 2),
                                             ("vio_partial", 4, 0), ("vio_partial", 8, 0), ("vio_13_frames_global_matrix", 2, 0)])
def test_sharded_solve_matches_oracle(oracle, case, world, mode):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hipemu"), "libpvio_hipemu.so"])
    pb = ba_compare.make(oracle, **ba_compare.CASES[case])
    st0, sm0 = BAState(pb), BASummary(pb)
    oracle.solve(pb, st0, sm0)
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "multi_rank_worker.py"), d, case, str(mode)]
        env = dict(os.environ, OMP_NUM_THREADS="1")
        subprocess.run(cmd, check=True, timeout=600, env=env, capture_output=True)
        rho = np.zeros(pb.n_landmarks)
        for r in range(world):
            z = np.load(os.path.join(d, "rank%d.npz" % r))
            assert int(z["iters"]) == sm0.num_iterations and int(z["term"]) == sm0.termination
            assert (z["succ"] == np.array([t["step_is_successful"] for t in sm0.trace()])).all()
            np.testing.assert_allclose(z["costs"], [t["cost"] for t in sm0.trace()], rtol=1e-7)
            # the landmark part of the gradient maximum crosses the ranks as one slot per rank inside the summing all-reduce
            np.testing.assert_allclose(z["gmax"], [t["gradient_max_norm"] for t in sm0.trace()], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(z["frame_state"], st0.frame_state, rtol=0, atol=1e-6)  # identical on every rank
            rho[int(z["l0"]):int(z["l1"])] = z["rho"]
        np.testing.assert_allclose(rho, st0.lm_inv_depth, rtol=0, atol=1e-6)
        if pb.use_inertial:  # the prior every rank builds from its shard + the all-reduce = the unsharded one
            _, _, IM0, iv0 = oracle.marginalize(pb, st0, 0)
            for r in range(world):
                z = np.load(os.path.join(d, "rank%d.npz" % r))
                np.testing.assert_allclose(z["marg_IM"], IM0, rtol=1e-5, atol=1e-7 * np.abs(IM0).max())
                np.testing.assert_allclose(z["marg_iv"], iv0, rtol=1e-5, atol=1e-6 * np.abs(iv0).max())
