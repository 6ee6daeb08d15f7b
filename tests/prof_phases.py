import sys, json; sys.path.insert(0,'.')
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate
for vio in (True, False):
    pb = synth.make_window(n_frames=10, n_landmarks=1000, use_inertial=vio, preintegrate=preintegrate if vio else None)
    ctx = HipContext(device=0); ctx.upload(pb)
    for _ in range(3): ctx.solve_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False)); prof = ctx.profile_resident(BASummary(pb, trace=False))
    print('vio' if vio else 'vision', {k:(round(v[0]/max(v[1],1)*1e3,1)) for k,v in prof.items()})
    for k in ('k_linearize','k_dense'):
        t = ctx.last_phase_ticks[k]; base=t[0]
        wall = (t[31]-t[30])*10.0  # ns at 100 MHz
        st = [x-base for x in t[:28]]
        print(' ', k, 'stamps(ticks)', st, 'wall_ns', wall, 'ticks/us', (max(st)/ (wall/1e3)) if wall>0 else None)
    ctx.close()
