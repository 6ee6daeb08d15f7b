"""Same-box A/B of the large-window role's LDS row padding (ba_types.h PVBA_TP_PAD): iterations/s and k_linearize per launch of the padded build (the product)
and of tests/micro/variants/tp_nopad.so (-DPVBA_TP_PAD=0), alternating, each in its own process.  Final costs are printed: the two layouts compute the same sums."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate
for n, m, vio in ((10, 50000, True), (30, 50000, True), (30, 50000, False), (10, 10000, True), (20, 20000, True)):
    pb = synth.make_window(n_frames=n, n_landmarks=m, use_inertial=vio, preintegrate=preintegrate if vio else None)
    ctx = HipContext(device=0, linearize_mode=2)
    ctx.upload(pb)
    sm = BASummary(pb, trace=False)
    for _ in range(3):
        ctx.solve_resident(sm)
    t0, it = time.perf_counter(), 0
    for _ in range(8):
        ctx.solve_resident(sm)
        it += sm.num_iterations
    rate = it / (time.perf_counter() - t0)
    ctx.profile_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False))
    print("  %%2d x %%5d %%-6s %%7.0f it/s  k_linearize %%6.1f us  final cost %%.12e" %% (n, m, "vio" if vio else "vision", rate, prof["k_linearize"][0] / max(prof["k_linearize"][1], 1) * 1e3, sm.final_cost), flush=True)
    ctx.close()
''' % ROOT
for rnd in range(2):
    for name, lib in (("unpadded", os.path.join(ROOT, "tests/micro/variants/tp_nopad.so")), ("padded (product)", None)):
        env = dict(os.environ)
        if lib:
            env["PVIO_HIP_LIB"] = lib
        else:
            env.pop("PVIO_HIP_LIB", None)
        print("%s, pass %d" % (name, rnd), flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
