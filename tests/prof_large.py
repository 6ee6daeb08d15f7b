"""Phase stamps of k_linearize on a large window (block 0 = a landmark workgroup when there are no IMU factors)."""
import sys; sys.path.insert(0, '.')
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate
cfgs = [(30, 50000, False), (30, 50000, True), (10, 50000, False)]
for n, m, vio in cfgs:
    pb = synth.make_window(n_frames=n, n_landmarks=m, use_inertial=vio, preintegrate=preintegrate if vio else None)
    ctx = HipContext(device=0); ctx.upload(pb)
    for _ in range(2): ctx.solve_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False)); prof = ctx.profile_resident(BASummary(pb, trace=False))
    print(n, m, 'vio' if vio else 'vision', {k: (round(v[0] / max(v[1], 1) * 1e3, 1)) for k, v in prof.items()}, flush=True)
    t = ctx.last_phase_ticks['k_linearize']; base = t[0]
    print('  stamps', [x - base for x in t[:10]], [x - base for x in t[14:18]], 'wall_ns', (t[31] - t[30]) * 10.0, flush=True)
    ctx.close()
