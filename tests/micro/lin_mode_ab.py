"""Where the large-window landmark role (linearize_mode 2, csrc/ba_lin_tp.h) overtakes the register-tile role (mode 1): k_linearize per launch (hipEvents, eager) and
iterations/s (graph replays) of both on windows of growing size, one box."""
import sys, time
sys.path.insert(0, '.')
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate

cases = [(10, 1000, True), (10, 2000, True), (10, 3000, True), (10, 5000, True), (10, 10000, True), (10, 20000, True), (6, 3000, True), (6, 8000, True), (16, 3000, True), (16, 8000, True), (24, 3000, True), (24, 8000, True), (10, 5000, False)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split("x")) + (True,) for a in sys.argv[1:]]
for n, m, vio in cases:
    pb = synth.make_window(n_frames=n, n_landmarks=m, use_inertial=vio, preintegrate=preintegrate if vio else None)
    row = []
    for mode in (1, 2):
        ctx = HipContext(device=0, linearize_mode=mode)
        ctx.upload(pb)
        sm = BASummary(pb, trace=False)
        for _ in range(3):
            ctx.solve_resident(sm)
        t0, it = time.perf_counter(), 0
        for _ in range(10):
            ctx.solve_resident(sm)
            it += sm.num_iterations
        rate = it / (time.perf_counter() - t0)
        prof = ctx.profile_resident(BASummary(pb, trace=False))
        prof = ctx.profile_resident(BASummary(pb, trace=False))
        row.append((rate, prof["k_linearize"][0] / max(prof["k_linearize"][1], 1) * 1e3, sm.final_cost))
        ctx.close()
    print("%2d x %5d %s: register tiles %7.0f it/s (k_linearize %6.1f us) | large-window role %7.0f it/s (k_linearize %6.1f us)   final costs %.9e / %.9e" % (
        n, m, "vio" if vio else "vision", row[0][0], row[0][1], row[1][0], row[1][1], row[0][2], row[1][2]), flush=True)
