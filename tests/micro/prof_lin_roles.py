"""k_linearize, block 0 of the launch: ticks at `prologue done` (site 1) and at the end of its role (site 9), one run per site.  In a VIO window block 0 is the
first IMU workgroup (launch order [IMU | prior | planes | landmarks]), in a vision-only window a landmark workgroup: how long the two kinds of role take.
usage (GPU box): python tests/micro/prof_lin_roles.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pvio_amd import synth, BASummary  # noqa: E402
from pvio_amd.solver import HipContext, preintegrate  # noqa: E402

for vio in (True, False):
    pb = synth.make_window(n_frames=10, n_landmarks=1000, use_inertial=vio, preintegrate=preintegrate if vio else None)
    out = {}
    for sel in (1, 9):
        os.environ["PVIO_HIP_STAMP_SEL"] = str(sel)
        os.environ["PVIO_HIP_PROFILE_GRAPH"] = "1"
        ctx = HipContext(device=0)
        ctx.upload(pb)
        for _ in range(3):
            ctx.solve_resident(BASummary(pb, trace=False))
        vals = []
        for _ in range(3):
            ctx.profile_resident(BASummary(pb, trace=False))
            vals.append(ctx.last_phase_ticks["k_linearize"][sel])
        ctx.close()
        out[sel] = sorted(vals)[1]
    print("%s window, block 0 (%s workgroup): prologue done %d ticks, role ends %d ticks (role %d ticks = %.1f us at 2.4 GHz)" % (
        "VIO" if vio else "vision-only", "IMU" if vio else "landmark", out[1], out[9], out[9] - out[1], (out[9] - out[1]) / 2400.0))
