"""k_linearize, VIO window: how long the first IMU workgroup (block 0) and the first prior workgroup run (shader-clock ticks),
next to the phases of a landmark workgroup (vision window, block 0)."""
import os, sys
sys.path.insert(0, '.')
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate
os.environ["PVIO_HIP_STAMP_SEL"] = "-1"
os.environ["PVIO_HIP_PROFILE_GRAPH"] = "0"
for vio in (True, False):
    pb = synth.make_window(n_frames=10, n_landmarks=1000, use_inertial=vio, preintegrate=preintegrate if vio else None)
    ctx = HipContext(device=0)
    ctx.upload(pb)
    for _ in range(3):
        ctx.solve_resident(BASummary(pb, trace=False))
    ctx.profile_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False))
    t = ctx.last_phase_ticks["k_linearize"]
    print("vio" if vio else "vision", "block 0 stamps (ticks since launch):", [int(x) for x in t[:10]])
    if vio:
        print("   first IMU workgroup: %d ticks   first prior workgroup: %d ticks   (role bodies, after the prologue)" % (t[11] - t[10], t[13] - t[12]))
    print("   wall clock of block 0: %.2f us; per-launch event times: %s" % ((t[31] - t[30]) * 0.01, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()}))
    ctx.close()
