"""One rank of the CPU multi-process test of the landmark-sharded solve (launched by tests/test_multi_rank_cpu.py with
torch.distributed.run, backend gloo).  Kernels run in the fiber emulator; the RCCL all-reduce is replaced by gloo."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir = sys.argv[1]
    case = sys.argv[2]
    linearize_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import ba_compare
    from oracle import oracle_py as O
    from pvio_amd import BAState, BASummary, capi
    from pvio_amd.solver import HipContext

    lib = capi.load(os.path.join(ROOT, "tests", "hipemu", "libpvio_hipemu.so"))

    @C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_long, C.c_int)
    def allreduce(buf, n, op_max):
        a = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op_max else dist.ReduceOp.SUM)
        return 0

    lib.hipemu_set_allreduce(allreduce)
    graph = case.endswith("@graph")  # resident re-solves through the slot GRAPH: both collectives of an iteration captured in it, > 1 rank
    timeout = case.endswith("@timeout")  # a time limit that only ONE rank's clock exceeds (PVIO_HIP_DEBUG_TIMEOUT_RANK): nobody may hang
    case = case.split("@")[0]
    pb = ba_compare.make(O, **{**ba_compare.CASES, **ba_compare.BIG_CASES}[case])
    if timeout:
        pb.max_solver_time = 100.0  # a real-time style limit: the clock is looked at every two slots
    shard = pb.shard(rank, world)
    ctx = HipContext(lib=lib, rank=rank, world_size=world, use_graph=graph, linearize_mode=linearize_mode)
    uid = (C.c_uint8 * 128)()
    assert lib.pvio_hip_comm_unique_id(uid) == 0
    assert lib.pvio_hip_comm_init(ctx.ctx, uid, rank, world) == 0
    st, sm = ctx.solve(shard)
    l0, l1 = shard.meta["lm_range"] if world > 1 else (0, pb.n_landmarks)
    graph_info = {}
    if graph:
        # the first solve of an upload launches eagerly; the next ones capture the slot graph (kernels + the two all-reduces per iteration) and
        # replay it.  Every replay must reproduce the eager solve bit for bit on every rank.
        ctx.upload(shard)
        outs = []
        for _ in range(3):
            smr = BASummary(shard, trace=False)
            ctx.solve_resident(smr)
            str_ = BAState(shard)
            ctx.download(str_)
            outs.append((str_.frame_state.copy(), str_.lm_inv_depth.copy(), smr.num_iterations))
        graph_info = dict(graph_same=np.array([int((o[0] == outs[0][0]).all() and (o[1] == outs[0][1]).all() and o[2] == outs[0][2]) for o in outs]),
                          graph_vs_eager=np.array([np.abs(outs[-1][0] - st.frame_state).max(), np.abs(outs[-1][1] - st.lm_inv_depth).max() if len(st.lm_inv_depth) else 0.0]),
                          graph_replays=np.array([ctx.graph_replays()]))
    # marginalize_frame on the sharded window: every rank sums its landmarks' part, the reduced buffer is all-reduced
    marg = {}
    if pb.use_inertial and not timeout and not graph:
        S, s_, IM, iv = ctx.marginalize(shard, st, 0)
        marg = dict(marg_IM=IM, marg_iv=iv)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), frame_state=st.frame_state, rho=st.lm_inv_depth, l0=l0, l1=l1,
             iters=sm.num_iterations, term=sm.termination, costs=np.array([t["cost"] for t in sm.trace()]),
             succ=np.array([t["step_is_successful"] for t in sm.trace()]), gmax=np.array([t["gradient_max_norm"] for t in sm.trace()]), **marg, **graph_info)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
