"""Cost of one complete keyframe solve as the tracker issues it (upload the window, solve, read the states back) against the
resident solve bench.py times."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from pvio_amd import synth, BAState, BASummary
from pvio_amd.solver import HipContext, preintegrate
for (n, m) in ((10, 1000), (10, 200), (30, 50000)):
    pb = synth.make_window(n_frames=n, n_landmarks=m, use_inertial=True, preintegrate=preintegrate)
    ctx = HipContext(device=0)
    for _ in range(3):
        ctx.solve(pb, trace=False)
    reps = 30 if m < 10000 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        st, sm = ctx.solve(pb, trace=False)
    full = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.upload(pb)
    up = (time.perf_counter() - t0) / reps
    sm = BASummary(pb, trace=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.solve_resident(sm)
    res = (time.perf_counter() - t0) / reps
    print('%dx%d: full solve %.3f ms  upload %.3f ms  resident solve %.3f ms (%d iterations)' % (n, m, 1e3 * full, 1e3 * up, 1e3 * res, sm.num_iterations), flush=True)
    ctx.close()
# a tracker never solves the same window twice: alternate two windows of different sizes
pbs = [synth.make_window(n_frames=10, n_landmarks=m, use_inertial=True, preintegrate=preintegrate) for m in (1000, 960)]
for graph in (True, False):
    ctx = HipContext(device=0, use_graph=graph)
    for k in range(6):
        ctx.solve(pbs[k & 1], trace=False)
    t0 = time.perf_counter()
    for k in range(40):
        st, sm = ctx.solve(pbs[k & 1], trace=False)
    print('alternating 10x1000 / 10x960 windows, graph=%s: full solve %.3f ms' % (graph, 1e3 * (time.perf_counter() - t0) / 40), flush=True)
    ctx.close()
