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
                                             ("vio_partial", 4, 0), ("vio_partial", 8, 0), ("vio_13_frames_global_matrix", 2, 0),
                                             ("vio_duplicate_blocks", 2, 2),
                                             # round 6: ranks WITHOUT a landmark (5 landmarks over 8 ranks), both landmark roles
                                             ("vio_five_landmarks", 8, 0), ("vio_five_landmarks", 8, 2),
                                             # round 3 (the emulator got fast enough): the metric window on four ranks, rejected steps /
                                             # the bias quirk / rotation priors on shards, the 20-frame HBM-matrix window on three
                                             ("metric_10x1000_vio", 4, 0), ("vio_zero_bias_quirk", 2, 0), ("config1_10x200", 3, 0),
                                             ("vio_20_frames_panels", 3, 0), ("vio_rot_prior", 2, 0)])
def test_sharded_solve_matches_oracle(oracle, case, world, mode):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hipemu"), "libpvio_hipemu.so"])
    pb = ba_compare.make(oracle, **{**ba_compare.CASES, **ba_compare.BIG_CASES}[case])
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


# (mode 2: the large-window landmark role on the shards -- what bench.py's sharded 10 KF x 50 000 leg runs inside the captured graph)
@pytest.mark.parametrize("case,world,mode", [("vio_partial", 2, 0), ("vio_plane", 4, 0), ("config1_10x200", 8, 0), ("vio_partial", 2, 2), ("config1_10x200", 4, 2)])
def test_sharded_graph_replay_with_captured_collectives(oracle, case, world, mode):
    """the landmark-sharded iteration inside the slot GRAPH -- kernels and both all-reduces captured (the emulator's stream
    capture records the collective as a node, as RCCL's does) -- replayed with more than one rank: every replay equals the eager solve of the
    same shards bit for bit on every rank, and the oracle within 1e-6."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hipemu"), "libpvio_hipemu.so"])
    pb = ba_compare.make(oracle, **{**ba_compare.CASES, **ba_compare.BIG_CASES}[case])
    st0, sm0 = BAState(pb), BASummary(pb)
    oracle.solve(pb, st0, sm0)
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + ((os.getpid() + 13 * world) % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "multi_rank_worker.py"), d, case + "@graph", str(mode)]
        subprocess.run(cmd, check=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True)
        for r in range(world):
            z = np.load(os.path.join(d, "rank%d.npz" % r))
            assert int(z["iters"]) == sm0.num_iterations and int(z["term"]) == sm0.termination
            assert int(z["graph_replays"][0]) >= 2, z["graph_replays"]  # the re-solves really went through the captured graph (no silent eager fallback)
            assert z["graph_same"].all(), z["graph_same"]              # three replays, identical
            assert (z["graph_vs_eager"] == 0).all(), z["graph_vs_eager"]  # and identical to the eager solve
            np.testing.assert_allclose(z["frame_state"], st0.frame_state, rtol=0, atol=1e-6)


@pytest.mark.parametrize("slow_rank", [0, 1])
def test_sharded_time_limit_is_agreed_across_ranks(oracle, slow_rank):
    """max_solver_time in a landmark-sharded solve: only ONE rank's clock runs out (PVIO_HIP_DEBUG_TIMEOUT_RANK); the
    decision is all-reduced, so both ranks stop in the same round -- with the same iteration count, state and NO_CONVERGENCE -- instead
    of one rank leaving and the other waiting in the next replay's all-reduce for good (the worker would hit the 300 s timeout)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hipemu"), "libpvio_hipemu.so"])
    case, world = "vio_partial", 2
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + ((os.getpid() + 7 + slow_rank) % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "multi_rank_worker.py"), d, case + "@timeout", "0"]
        env = dict(os.environ, OMP_NUM_THREADS="1", PVIO_HIP_DEBUG_TIMEOUT_RANK=str(slow_rank))
        subprocess.run(cmd, check=True, timeout=300, env=env, capture_output=True)
        z = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
        assert int(z[0]["iters"]) == int(z[1]["iters"]) and int(z[0]["term"]) == int(z[1]["term"]) == 1  # NO_CONVERGENCE on both
        assert 1 <= int(z[0]["iters"]) <= 2  # the clock is looked at after the first two slots (iteration 0 + one step)
        np.testing.assert_array_equal(z[0]["frame_state"], z[1]["frame_state"])


def test_time_limit_single_gpu_stops_between_chunks(oracle):
    """a real-time limit on one GPU (emulated): the slot graph is cut into two-slot chunks and the clock is looked at between them"""
    from pvio_amd import capi
    from pvio_amd.solver import HipContext
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hipemu"), "libpvio_hipemu.so"])
    lib = capi.load(os.path.join(ROOT, "tests", "hipemu", "libpvio_hipemu.so"))
    ctx = HipContext(lib=lib, use_graph=True)
    pb = ba_compare.make(oracle, **ba_compare.CASES["vio_partial"])
    pb.max_solver_time = 1e-9
    st, sm = ctx.solve(pb)
    assert sm.termination == 1 and sm.is_usable and 1 <= sm.num_iterations <= 2
    pb.max_solver_time = 1.0e6
    st, sm = ctx.solve(pb)
    assert sm.num_iterations == 10
    ctx.close()
