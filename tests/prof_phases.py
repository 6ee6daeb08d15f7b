"""Phase profile of the four kernels of a bundle-adjustment iteration (10 x 1000, VIO and vision-only).
Per-kernel times: hipEvents around every launch of the profiling entry point, WITHOUT stamps (PVIO_HIP_STAMP_SEL=-2).
Phases: one run per stamp site (PVIO_HIP_STAMP_SEL=k: only site k stores, ticks since the start of its own launch), so that
the stamps' own waits and stores do not add up; `all` = every site active in one run, for comparison (that kernel is slower).
Since round 4 the per-panel sites of k_dense's look-ahead loop (8-17) are compiled in only with -DPVIO_DENSE_LOOP_STAMPS: for those rows run
`python tests/micro/build_variant.py loop_stamps` first and `PVIO_HIP_LIB=tests/micro/variants/loop_stamps.so python tests/prof_phases.py` (zeros otherwise)."""
import os, sys
sys.path.insert(0, '.')
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate

DENSE_SITES = [(28, "round 1 of requests back"), (29, "control inputs in LDS"), (1, "control done"), (2, "vectors assembled"), (3, "finalize done"), (4, "scaled vectors"), (18, "wave 1: last tile request back"), (19, "wave 1: tiles scaled"), (8, "tiles scaled, first panel starts"),
               (10, "panel 0: L rows stored"), (11, "panel 0: update done"), (12, "panel 0: end"), (13, "panel 10 starts"), (14, "panel 10: pivot loop done"), (15, "panel 10: L rows stored"), (17, "panel 10: end"),
               (5, "factorization done"), (6, "back substitution done"), (7, "end")]
LIN_SITES = [(1, "prologue done"), (2, "first chunk: cleared"), (3, "factors evaluated"), (4, "landmark sums"), (5, "landmark scalars"), (6, "tile accumulation starts"),
             (7, "tiles accumulated"), (8, "partial row flushed"), (9, "end")]


def run(pb, sel, graph=False):
    os.environ["PVIO_HIP_STAMP_SEL"] = str(sel)
    os.environ["PVIO_HIP_PROFILE_GRAPH"] = "1" if graph else "0"
    ctx = HipContext(device=0)
    ctx.upload(pb)
    for _ in range(3):
        ctx.solve_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False))
    ticks = {k: list(v) for k, v in ctx.last_phase_ticks.items()} if sel != -2 else None
    ctx.close()
    return prof, ticks


for vio in (True, False):
    pb = synth.make_window(n_frames=10, n_landmarks=1000, use_inertial=vio, preintegrate=preintegrate if vio else None)
    prof, _ = run(pb, -2)
    print('vio' if vio else 'vision', 'per launch, no stamps (us):', {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()},
          'launches per solve:', {k: v[1] for k, v in prof.items()}, flush=True)
    ctx = HipContext(device=0)
    ctx.upload(pb)
    sm = BASummary(pb, trace=False)
    for _ in range(20):
        ctx.solve_resident(sm)
    import time
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.solve_resident(sm)
    print('  graph replay: %.1f us per solve, %d iterations, termination %d' % ((time.perf_counter() - t0) * 1e4, sm.num_iterations, sm.termination), flush=True)
    ctx.close()
    _, allt = run(pb, -1)
    for kern, sites, block0 in (("k_dense", DENSE_SITES, True), ("k_linearize", LIN_SITES, not vio)):
        if not block0:
            continue  # block 0 of the VIO grid is an IMU workgroup: the landmark phases are stamped in the vision-only run
        row = []
        for idx, name in sites:
            _, t = run(pb, idx)
            _, tg = run(pb, idx, graph=True)
            row.append((name, t[kern][idx], allt[kern][idx], tg[kern][idx]))
        print('  %s: ticks since the launch started.  eager launches: one site per run | all sites in one run || inside a graph replay, one site per run' % kern)
        for name, one, al, gr in row:
            print('    %-36s %8d | %8d || %8d' % (name, one, al, gr))
        wall_us = (allt[kern][31] - allt[kern][30]) * 0.01
        print('    wall clock of the last launch that reached the end (a factoring launch for k_dense; all sites active): %.1f us -> %.0f ticks/us'
              % (wall_us, row[-1][2] / wall_us if wall_us > 0 else 0))
