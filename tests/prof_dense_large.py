"""Phase stamps of k_dense when the reduced system does not fit the register / LDS path (P > 168)."""
import sys; sys.path.insert(0, '.')
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate
for n, m, vio in [(30, 3000, True), (30, 3000, False), (20, 3000, True), (13, 3000, True)]:
    pb = synth.make_window(n_frames=n, n_landmarks=m, use_inertial=vio, preintegrate=preintegrate if vio else None)
    ctx = HipContext(device=0); ctx.upload(pb)
    for _ in range(2): ctx.solve_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False)); prof = ctx.profile_resident(BASummary(pb, trace=False))
    print(n, m, 'vio' if vio else 'vision', {k: (round(v[0] / max(v[1], 1) * 1e3, 1)) for k, v in prof.items()}, flush=True)
    t = ctx.last_phase_ticks['k_dense']; base = t[0]
    print('  stamps', {i: x - base for i, x in enumerate(t[:28]) if x != 0}, 'wall_ns', (t[31] - t[30]) * 10.0, flush=True)
    ctx.close()
