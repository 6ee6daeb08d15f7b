"""Up to which landmark count the back-substitution should stay inside k_dense (Dims::fuse_backsub; PVIO_HIP_FUSE_BACKSUB_MAX): iterations/s of windows of the
reference's own sizes with the threshold at 256 (shipped), 512 and 1024, each setting in its own process, alternating."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from pvio_amd import synth, BASummary
from pvio_amd.solver import HipContext, preintegrate
for n, m, vis, vio in ((8, 200, 5, True), (8, 300, 5, True), (8, 400, 5, True), (8, 500, 4, True), (10, 600, 6, True), (8, 800, 5, True), (10, 1000, None, True), (8, 300, 5, False)):
    pb = synth.make_window(n_frames=n, n_landmarks=m, use_inertial=vio, visibility=vis, preintegrate=preintegrate if vio else None)
    ctx = HipContext(device=0)
    ctx.upload(pb)
    sm = BASummary(pb, trace=False)
    for _ in range(10):
        ctx.solve_resident(sm)
    t0, it = time.perf_counter(), 0
    for _ in range(100):
        ctx.solve_resident(sm)
        it += sm.num_iterations
    print("  %%2d x %%4d %%-6s %%7.0f it/s  final cost %%.12e" %% (n, m, "vio" if vio else "vision", it / (time.perf_counter() - t0), sm.final_cost), flush=True)
    ctx.close()
''' % ROOT
for rnd in range(2):
    for thr in (256, 512, 1024):
        env = dict(os.environ, PVIO_HIP_FUSE_BACKSUB_MAX=str(thr))
        print("fused up to %d landmarks, pass %d" % (thr, rnd), flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
