"""The window the reference actually solves (config/euroc.yaml: sliding_window_size 8; a few hundred landmarks seen by about half of the frames): per-kernel times,
graph-replay rate and the role times of k_linearize (first IMU / prior workgroup, a landmark workgroup), next to the metric window."""
import os, sys, time
sys.path.insert(0, '.')
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate
os.environ["PVIO_HIP_STAMP_SEL"] = "-1"
os.environ["PVIO_HIP_PROFILE_GRAPH"] = "0"
for n, m, vis in ((8, 300, 5), (8, 150, 4), (9, 300, 5), (10, 1000, None)):
    for vio in (True, False):
        pb = synth.make_window(n_frames=n, n_landmarks=m, use_inertial=vio, visibility=vis, preintegrate=preintegrate if vio else None)
        ctx = HipContext(device=0)
        ctx.upload(pb)
        sm = BASummary(pb, trace=False)
        for _ in range(5):
            ctx.solve_resident(sm)
        t0, it, slots = time.perf_counter(), 0, 0
        for _ in range(50):
            ctx.solve_resident(sm)
            it += sm.num_iterations
        dt = time.perf_counter() - t0
        ctx.profile_resident(BASummary(pb, trace=False))
        prof = ctx.profile_resident(BASummary(pb, trace=False))
        t = ctx.last_phase_ticks["k_linearize"]
        print("%2d x %4d %-6s vis %s: %6.0f it/s, %.0f us per solve (%d iterations); per launch (eager, us): %s" % (
            n, m, "vio" if vio else "vision", vis, it / dt, dt / 50 * 1e6, sm.num_iterations, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()}), flush=True)
        if vio:
            print("      block 0 = first IMU workgroup: prologue %d ticks, role %d ticks; first prior workgroup %d ticks; block 0 wall %.1f us" % (t[1], t[11] - t[10], t[13] - t[12], (t[31] - t[30]) * 0.01))
        else:
            print("      block 0 = a landmark workgroup: stamps %s; wall %.1f us" % ([int(x) for x in t[:10]], (t[31] - t[30]) * 0.01))
        ctx.close()
