"""Phase stamps of the large-window landmark role (ba_lin_tp.h) -- block 0 is a landmark workgroup in a vision-only window.  Sites: 1 prologue done,
2 last chunk starts, 3 E (factors evaluated), 4 D (direct part), 5 L (per-landmark Gram) + barrier, 6 P (scalars), 7 S (Schur + next chunk's inputs),
8 flushed, 9 end.  Ticks of the shader clock since the launch started; all sites active (the kernel runs a little slower)."""
import os, sys
sys.path.insert(0, '.')
os.environ["PVIO_HIP_STAMP_SEL"] = "-1"
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate
cfgs = [(10, 50000), (30, 50000), (10, 10000)]
for n, m in cfgs:
    pb = synth.make_window(n_frames=n, n_landmarks=m, use_inertial=False)
    ctx = HipContext(device=0); ctx.upload(pb)
    for _ in range(2): ctx.solve_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False)); prof = ctx.profile_resident(BASummary(pb, trace=False))
    print(n, m, 'vision', {k: (round(v[0] / max(v[1], 1) * 1e3, 1)) for k, v in prof.items()}, flush=True)
    t = ctx.last_phase_ticks['k_linearize']
    wall_us = (t[31] - t[30]) * 0.01
    print('  stamps (ticks since launch): prologue %d | last chunk start %d  E %d  D %d  L %d  P %d  S %d | flushed %d  end %d   wall %.1f us -> %.0f ticks/us' % (
        t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], wall_us, t[9] / wall_us if wall_us > 0 else 0), flush=True)
    print('  last chunk: E %d  D %d  L %d  P %d  S %d ticks;  chunks of workgroup 0: (start of last chunk - prologue) / chunk = ?' % (t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6]))
    print('  flush detail: table requested %d  sums stored in LDS (2 barriers) %d  subs added %d  DS written %d' % (t[18] - t[7], t[20] - t[18], t[19] - t[20], t[14] - t[19]))
    if t[21]:
        print('  flush straight to the row (many frames): vectors + row zeroed %d  tiles scattered %d  direct blocks %d  anchor block + scalars %d' % (t[21] - t[14], t[22] - t[21], t[23] - t[22], t[8] - t[23]))
    print('  flush: task sums in LDS %d | first pass: tiles scattered %d  direct blocks added %d  row written %d | flushed %d' % (t[14] - t[7], t[15] - t[14], t[16] - t[15], t[17] - t[16], t[8] - t[7]))
    ctx.close()
