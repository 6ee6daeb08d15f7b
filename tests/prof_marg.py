"""Time of pvio_hip_ba_marginalize (upload + MODE_MARG linearization + reduction + Schur complement + eigen-decomposition + read-back)
on the metric's window, next to a complete solve of the same window."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from pvio_amd import synth, BAState
from pvio_amd.solver import HipContext, preintegrate
import marg_compare
from oracle import oracle_py as O
O.build()
ctx = HipContext(device=0)
for nf, nl in ((10, 1000), (10, 200), (8, 300)):
    pb, st = marg_compare.solved_window(O, regular_prior=False, n_frames=nf, n_landmarks=nl, use_inertial=True)
    for _ in range(5):
        ctx.marginalize(pb, st, 0, want_info=False)
    if len(sys.argv) > 1:  # dump the Schur complement for tests/micro/eig_bench.cpp
        IM = ctx.marginalize(pb, st, 0)[2]
        with open("%s_%dx%d.bin" % (sys.argv[1], nf, nl), "wb") as f:
            f.write(np.int32(IM.shape[0]).tobytes()); f.write(np.ascontiguousarray(IM).tobytes())
    t0 = time.perf_counter()
    for _ in range(30):
        ctx.marginalize(pb, st, 0, want_info=False)
    t_m = (time.perf_counter() - t0) / 30
    for _ in range(3):
        ctx.solve(pb, trace=False)
    t0 = time.perf_counter()
    for _ in range(30):
        ctx.solve(pb, trace=False)
    t_s = (time.perf_counter() - t0) / 30
    print("%dx%d: marginalize_frame(0) %.3f ms   complete solve %.3f ms" % (nf, nl, 1e3 * t_m, 1e3 * t_s))
