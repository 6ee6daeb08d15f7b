"""A/B timing of two builds of libpvio_hip.so on the SAME box (the pool varies by +-4 % from box to box): resident solves of the
10 x 1000 VIO window (or `frames landmarks` given after the two paths), alternating between the libraries.
usage: python tests/prof_ab.py libA.so libB.so [frames landmarks]"""
import sys, time
sys.path.insert(0, '.')
from pvio_amd import synth, BASummary, capi
from pvio_amd.solver import HipContext, preintegrate

paths = sys.argv[1:3]
nf, nl = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (10, 1000)
reps = 300 if nl <= 2000 else 12
pb = synth.make_window(n_frames=nf, n_landmarks=nl, use_inertial=True, preintegrate=preintegrate)
ctxs = []
for p in paths:
    ctx = HipContext(lib=capi.load(p), device=0)
    ctx.upload(pb)
    for _ in range(20 if nl <= 2000 else 3):
        ctx.solve_resident(BASummary(pb, trace=False))
    ctxs.append(ctx)
res = [[] for _ in paths]
for rnd in range(6):
    for i, ctx in enumerate(ctxs):
        sm = BASummary(pb, trace=False)
        t0 = time.perf_counter()
        its = 0
        for _ in range(reps):
            ctx.solve_resident(sm)
            its += sm.num_iterations
        dt = time.perf_counter() - t0
        res[i].append(its / dt)
for p, r in zip(paths, res):
    r = sorted(r)
    print("%-40s median %.0f it/s  (min %.0f max %.0f)" % (p, r[len(r) // 2], r[0], r[-1]))
