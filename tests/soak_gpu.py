"""Soak run on the GPU: thousands of complete keyframe solves over alternating window shapes with image pairs (undistortion,
pyramid, LK, detection) in between; device memory must not grow after the warm-up and nothing may hang.
Last run (round 4, profiles/r4_soak.txt): 3000 solves (30 000 iterations) + 300 image pairs in 3.2 s, 197.1 MB before and after."""
import sys, time; sys.path.insert(0, '.')
import torch
from pvio_amd import synth
from pvio_amd.solver import HipContext, HipImage, HipUndistort, klt_track, detect_corners, preintegrate
free0 = torch.cuda.mem_get_info(0)[0]
ctx = HipContext(device=0)
pbs = [synth.make_window(n_frames=n, n_landmarks=m, use_inertial=v, preintegrate=preintegrate if v else None)
       for n, m, v in ((10, 1000, True), (8, 700, False), (13, 400, True), (10, 960, True))]
# round 6: a window of 54 000 factors (the large-window landmark role, csrc/ba_lin_tp.h) every 50th solve, and LK launches of 150 tracks (k_lk_track_levels) beside the 1500
pb_large = synth.make_window(n_frames=10, n_landmarks=6000, use_inertial=True, preintegrate=preintegrate)
img0, img1, p, truth, init = synth.make_image_pair(512, 512, 1500)
ud = HipUndistort(ctx, *synth.make_undistort_maps(512, 512))


def pair():
    A, B = HipImage(ctx, img0, undistort=ud), HipImage(ctx, img1, undistort=ud)
    klt_track(ctx, A, B, p, init)
    klt_track(ctx, A, B, p[:150], init[:150])
    detect_corners(ctx, A)
    A.release(), B.release()


for pb in pbs + [pb_large]:
    ctx.solve(pb, trace=False)
pair()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info(0)[0]
t0, iters = time.perf_counter(), 0
for k in range(3000):
    st, sm = ctx.solve(pb_large if k % 50 == 49 else pbs[k % 4], trace=False)
    iters += sm.num_iterations
    if k % 10 == 0:
        pair()
torch.cuda.synchronize()
free2 = torch.cuda.mem_get_info(0)[0]
print('3000 keyframe solves over 5 window shapes (one of 54 000 factors every 50th) + 300 image pairs with 1500- and 150-track LK launches in %.1f s (%d iterations); device memory in use after warm-up %.1f MB, at the end %.1f MB'
      % (time.perf_counter() - t0, iters, (free0 - free1) / 1e6, (free0 - free2) / 1e6))
assert free2 >= free1 - (1 << 20), "device memory grew"
ctx.close()
