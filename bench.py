#!/usr/bin/env python3
"""bench.py -- BA iterations/sec of the hand-written HIP bundle adjustment on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is ONE full `BundleAdjustor::solve` of one synthetic sliding window (Ceres-1.14 Dogleg semantics, at most
`solver.iteration_limit` = 10 trust-region iterations, config/euroc.yaml:66) with the window already resident in HBM
(pvio_hip_ba_upload before the timed region).  value = trust-region iterations executed / wall time, the metric of
BASELINE.json.  Default workload = the configuration the metric is quoted on: 10 keyframes x 1000 landmarks, full
factor set of a steady-state VIO window (9000 reprojection factors with Cauchy loss, 9 IMU pre-integration factors,
the gauge/marginalization prior).  `--workload vision` gives the reprojection-only variant.

With N > 1 the window is landmark-sharded over the ranks (contiguous CSR ranges); the reduced pose system and 8 scalars are
all-reduced with RCCL once per linearization / back-substitution -> "scaling": "strong".  `value` is the metric's window
(10 KF x 1000 landmarks) at every N; the window north_star states the multi-GPU target on, 10 KF x 50 000 landmarks, is the
`scaling_window` object of every line (sharded like the headline; for N > 1 with the same window's unsharded rate on rank 0
from the same run and the ratio of the two).

Extra objects on the JSON line:
  roofline      the kernel that moves the window's data (k_linearize): algorithmic bytes per launch / its average
                duration, measured here with hipEvents on the solver's own stream (pvio_hip_ba_profile_resident), against the 8 TB/s
                HBM peak; `traffic` = HBM bytes per launch from rocprofv3 counter passes run as children of THIS run (null
                when the profiler is not there).
  roofline_dense  the kernel that is longest by duration (k_dense, one workgroup): flops of the Cholesky + substitutions per
                factoring launch / its duration against the FP64 peak.
  scaling_window  the 10 KF x 50 000 landmark VIO window (the one north_star states the 8-GPU target on), sharded the
                same way, a few solves: its iterations/s at this N next to the headline value.
  reference_window  the window the reference's own configuration solves (8 KF, 300 landmarks seen by 5 frames each, full VIO factor
                set): iterations/s resident, and the CPU restatement on the same window (2 s sample).
  api           the same window through pvio_hip_ba_solve (upload + iterations + download per solve): the H2D/D2H-inclusive rate.
  cpu_baseline  the CPU oracle (oracle/, a single-threaded restatement of the reference's Ceres path; the real
                reference cannot be built here) timed on the same window on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 (vector = matrix) peak: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz -- the rate of the measured
                         # 64-cycle v_mfma_f64_16x16x4_f64 (profiles/r2_ubench_mfma_valu_overlap.txt); the guide lists no FP64 figure

_PMC = {"done": False, "kernels": None, "note": None, "stats": None}


def live_pmc(args):
    """HBM bytes per launch and kernel, measured INSIDE this run: two child passes of this same script under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and WRITE_SIZE in separate passes, counters only -- no sys / runtime
    traces), reduced as /opt/skills/guides/MI355X_MICROARCH.md prescribes (profiles/summarize_pmc.py: KiB units, FETCH_SIZE doubled
    on gfx950).  Returns None (-> `traffic: null`) when rocprofv3 is missing, a pass fails or takes too long.  Nothing is read
    from files committed earlier: the numbers belong to the binaries that are being timed."""
    if _PMC["done"]:
        return _PMC["kernels"]
    _PMC["done"] = True
    import shutil
    import subprocess
    import tempfile
    if getattr(args, "no_pmc", False) or os.environ.get("PVIO_BENCH_NO_PMC") or shutil.which("rocprofv3") is None:
        _PMC["note"] = "rocprofv3 not run"
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import summarize_pmc
        res = {}
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(td, counter)
                cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                       os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", args.workload]
                env = dict(os.environ, TMPDIR="/tmp", PVIO_BENCH_NO_PMC="1")
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=90)
                if r.returncode != 0:
                    _PMC["note"] = "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
                    return None
                res[counter] = summarize_pmc.per_kernel(out, counter)
            # third pass, no counters: rocprofv3 --kernel-trace --stats of the same child -- per-kernel durations INSIDE hipGraph replays (the
            # hipEvent figures of `kernel_us` come from eager launches with a host synchronization per slot and run ~10 % longer)
            try:
                out = os.path.join(td, "stats")
                cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "--", sys.executable,
                       os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", args.workload]
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=90)
                if r.returncode == 0:
                    _PMC["stats"] = summarize_pmc.kernel_stats(out)
            except Exception:
                _PMC["stats"] = None
        kernels = {}
        for k in set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"]):
            f, nf = res["FETCH_SIZE"].get(k, (0.0, 0))
            w, nw = res["WRITE_SIZE"].get(k, (0.0, 0))
            kernels[k] = {"hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0, "launches": [nf, nw]}
        _PMC["kernels"] = kernels
        _PMC["note"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes of this run; bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction)"
        return kernels
    except Exception as e:  # a profiler that is absent or hangs must not cost the bench line
        _PMC["note"] = "pmc pass failed: %s" % type(e).__name__
        return None


def pmc_traffic(kernel, args=None):
    """`roofline.traffic` of `kernel`: live counters of this run (rank 0 of a single-GPU run) or None"""
    if args is None or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return None
    k = live_pmc(args)
    return None if not k or kernel not in k else k[kernel]["hbm_bytes_per_launch"]


def pmc_child(args):
    """What the counter passes profile: a few resident solves of the headline window and a few LK launches, nothing else."""
    from pvio_amd import BASummary, synth
    from pvio_amd.solver import HipContext, HipImage, klt_track, preintegrate
    pb, _ = build_window(args, preintegrate)
    ctx = HipContext(device=0)
    ctx.upload(pb)
    for _ in range(6):
        ctx.solve_resident(BASummary(pb, trace=False))
    img0, img1, p, truth, init = synth.make_image_pair(512, 512, 1500)
    A, B = HipImage(ctx, img0), HipImage(ctx, img1)
    for _ in range(6):
        klt_track(ctx, A, B, p, init)
    ctx.close()


def parse_workload(name):
    # "<frames>x<landmarks>[_vision|_vio]" or the shorthands "vio" / "vision"
    if name in ("vio", "vision"):
        return 10, 1000, name == "vio"
    body, _, kind = name.partition("_")
    n, m = body.lower().split("x")
    return int(n), int(m), (kind != "vision")


def build_window(args, preintegrate):
    from pvio_amd import synth
    n, m, vio = parse_workload(args.workload)
    return synth.make_window(n_frames=n, n_landmarks=m, use_inertial=vio, preintegrate=preintegrate if vio else None), (n, m, vio)


def bench_klt(ctx, args, width=512, height=512, n_points=1500, reps=50):
    """KLT tracks/ms on a TUM-VI-sized synthetic pair (BASELINE.json configs[3]: 512x512, 1500 tracks)."""
    from pvio_amd import synth
    from pvio_amd.solver import HipImage, HipUndistort, detect_corners, klt_track
    img0, img1, p, truth, init = synth.make_image_pair(width, height, n_points)
    A, B = HipImage(ctx, img0), HipImage(ctx, img1)   # upload + CLAHE + pyramid + Scharr on device
    # steady state of a camera stream: every new frame replaces the oldest pyramid (two stay alive, as in the tracker)
    live = [HipImage(ctx, img0), HipImage(ctx, img1)]
    t0 = time.perf_counter()
    for k in range(20):
        live.pop(0).release()
        live.append(HipImage(ctx, img1 if k & 1 else img0))
    prep_ms = 1e3 * (time.perf_counter() - t0) / 20
    for im in live:
        im.release()
    # the same stream with the dataset readers' undistortion in front (pvio_hip_image_create_undistorted: k_remap on the device)
    ud = HipUndistort(ctx, *synth.make_undistort_maps(width, height))
    live = [HipImage(ctx, img0, undistort=ud), HipImage(ctx, img1, undistort=ud)]
    t0 = time.perf_counter()
    for k in range(20):
        live.pop(0).release()
        live.append(HipImage(ctx, img1 if k & 1 else img0, undistort=ud))
    prep_ud_ms = 1e3 * (time.perf_counter() - t0) / 20
    for im in live:
        im.release()
    ud.release()
    # warm-up: the legs in front of this one are host work (pyramids through the wrapper, the CPU baseline of the BA leg), the GPU's clocks have
    # dropped; 5 launches were not enough to bring them back (the same launch read 36.9 us here and 30.9 us in a loop of its own on one box,
    # profiles/r6_klt_forms_check.txt)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.25:
        klt_track(ctx, A, B, p, init)
    dev_ms, t0 = 0.0, time.perf_counter()
    for _ in range(reps):
        q, st, ms = klt_track(ctx, A, B, p, init)
        dev_ms += ms
    wall_ms = 1e3 * (time.perf_counter() - t0) / reps
    dev_ms /= reps
    for _ in range(3):
        detect_corners(ctx, A)
    t0 = time.perf_counter()
    for _ in range(20):
        corners, _ = detect_corners(ctx, A)  # Harris response + NMS on the device, ordering + min-distance selection on the host
    detect_ms = 1e3 * (time.perf_counter() - t0) / 20
    # the outlier rejection of a frame: cv::findFundamentalMat(FM_RANSAC, 1 px, 0.99) over the tracked matches, hypotheses in batches on the device
    from pvio_amd.solver import fundamental_ransac
    # (matches of a 3-D scene seen by two cameras 0.3 m apart, 0.3 px of noise, every tenth match a gross outlier: the image pair above is a
    # plane under a homography, for which the fundamental matrix is not unique)
    rr = np.random.default_rng(8)
    Kc = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]])
    X3 = np.stack([rr.uniform(-3, 3, n_points), rr.uniform(-2, 2, n_points), rr.uniform(3, 9, n_points)], 1)
    ang = 0.06
    R2 = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    proj = lambda Xc: (Kc @ (Xc / Xc[:, 2:3]).T).T[:, :2]
    rp = (proj(X3) + rr.normal(0, 0.3, (n_points, 2))).astype(np.float32)
    rq = proj((R2 @ X3.T).T + np.array([0.3, 0.02, 0.05])) + rr.normal(0, 0.3, (n_points, 2))
    rq[::10] += rr.uniform(8, 60, (len(rq[::10]), 2)) * rr.choice([-1, 1], (len(rq[::10]), 2))
    rq = rq.astype(np.float32)
    for _ in range(3):
        fundamental_ransac(ctx, rp, rq)
    t0 = time.perf_counter()
    for _ in range(20):
        r_good, _, _, r_hyp = fundamental_ransac(ctx, rp, rq)
    ransac_ms = 1e3 * (time.perf_counter() - t0) / 20
    alg_bytes = 11616 * n_points  # 4 levels x (22^2 u8 template + 22^2 x 2 int16 derivatives + 22^2 u8 target)
    # the launch form (klt.hip, Klt::track): k_lk_track_levels (a workgroup per track, a wave per pyramid level) up to three tracks per CU
    # (the tracker's own regime: 150 keypoints), k_lk_track (a wave per track) beyond -- bench.py's 1500 tracks; PVIO_HIP_LK_FORM forces one
    try:
        import torch
        n_simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    except Exception:
        n_simds = 1024  # MI355X: 256 CUs
    form = os.environ.get("PVIO_HIP_LK_FORM", "0")
    lk_kernel = {"1": "k_lk_track", "2": "k_lk_track_units", "3": "k_lk_track_levels"}.get(form, "k_lk_track_levels" if 4 * n_points <= 3 * n_simds else "k_lk_track")
    out = {"metric": "KLT tracks/ms", "value": n_points / dev_ms, "unit": "tracks/ms", "value_incl_h2d_d2h": n_points / wall_ms,
           "workload": "%dx%d u8 pair, %d tracks, win 21x21, 4 levels, <=30 iterations, initial flow given" % (width, height, n_points),
           "tracked": int(st.sum()), "preprocess_ms_per_image": prep_ms, "preprocess_undistorted_ms_per_image": prep_ud_ms, "detect_ms_per_image": detect_ms, "detected_corners": int(len(corners)),
           "ransac_ms_per_frame": ransac_ms, "ransac": {"matches": int(len(rp)), "inliers": int(r_good), "hypotheses_evaluated": int(r_hyp),
                                                         "what": "pvio_hip_fundamental_ransac incl. the copies, 10 % gross outliers"},
           "roofline": {"bound": "hbm", "kernel": lk_kernel, "achieved": alg_bytes / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(lk_kernel, args), "algorithmic_bytes_per_launch": alg_bytes,
                        "avg_launch_us": dev_ms * 1e3}}
    if not args.no_cpu_baseline:
        from oracle import oracle_py as O
        P0, P1 = O.build_pyramid(O.clahe(img0)), O.build_pyramid(O.clahe(img1))
        t1, n = time.perf_counter(), 0
        while time.perf_counter() - t1 < min(args.cpu_seconds, 5.0):
            O.klt_track(P0, P1, p, init)
            n += 1
        cpu_ms = 1e3 * (time.perf_counter() - t1) / n
        out["cpu_baseline"] = {"value": n_points / cpu_ms, "unit": "tracks/ms", "cores": 1, "kind": "port",
                               "sample": "%d calls of the scalar restatement of calcOpticalFlowPyrLK on the same pair (OpenCV itself is "
                                         "parallel_for_ over points)" % n}
    A.release()
    B.release()
    return out


def bench_concurrent_windows(pb, streams=3, steps=60):
    """Aggregate iterations/s of `streams` independent windows solved at the same time (one context = one stream + one
    hipGraph each, one host thread each).  Not the headline metric (a VIO session solves one window at a time): it shows how
    much of the GPU the latency-bound single-window solve leaves free for other sessions."""
    import threading
    from pvio_amd import BASummary
    from pvio_amd.solver import HipContext
    ctxs = [HipContext(device=0) for _ in range(streams)]
    for c in ctxs:
        c.upload(pb)
        for _ in range(5):
            c.solve_resident(BASummary(pb, trace=False))
    iters = [0] * streams

    def work(i):
        sm = BASummary(pb, trace=False)
        for _ in range(steps):
            ctxs[i].solve_resident(sm)
            iters[i] += sm.num_iterations
    th = [threading.Thread(target=work, args=(i,)) for i in range(streams)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    return {"streams": streams, "value": sum(iters) / dt, "unit": "iterations/s (aggregate)", "steps_per_stream": steps}


def bench_reference_window(preintegrate, cpu_seconds=2.0, steps=100):
    """BA iterations/s of the window the reference itself solves in a session (config/euroc.yaml: sliding_window_size 8; a few
    hundred landmarks, each seen by about half of the frames), full VIO factor set, next to the CPU restatement timed on the same
    window.  Not the headline metric (north_star names the 10 KF x 1000 window): it says what a drop-in user of the reference's
    own configuration gets."""
    from pvio_amd import BAState, BASummary, synth
    from pvio_amd.solver import HipContext
    pb = synth.make_window(n_frames=8, n_landmarks=300, use_inertial=True, visibility=5, preintegrate=preintegrate)
    ctx = HipContext(device=0)
    ctx.upload(pb)
    sm = BASummary(pb, trace=False)
    for _ in range(10):
        ctx.solve_resident(sm)
    t0, it = time.perf_counter(), 0
    for _ in range(steps):
        ctx.solve_resident(sm)
        it += sm.num_iterations
    dt = time.perf_counter() - t0
    ctx.close()
    out = {"workload": "8 KF x 300 landmarks seen by 5 frames each, full VIO factor set, %d reprojection factors" % pb.n_obs,
           "value": it / dt, "unit": "iterations/s", "steps": steps, "us_per_solve": dt / steps * 1e6}
    if cpu_seconds > 0:
        from oracle import oracle_py as O
        O.build()
        c_it, c_t = 0, 0.0
        O.solve_fast(pb, BAState(pb), BASummary(pb, trace=False))
        while c_t < cpu_seconds:
            st, so = BAState(pb), BASummary(pb, trace=False)
            t1 = time.perf_counter()
            O.solve_fast(pb, st, so)
            c_t += time.perf_counter() - t1
            c_it += so.num_iterations
        out["cpu_baseline"] = {"value": c_it / c_t, "unit": "BA iterations/s", "cores": 1, "kind": "port", "sample": "%.1f s of solves of the same window" % c_t}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def bench_scaling_window(ctx, args, rank, world, dist, barrier, preintegrate, n_frames=10, n_landmarks=50000, steps=10, warmup=2):
    """BA iterations/s of the 10 KF x 50 000 landmark VIO window (BASELINE.json north_star: the window the 8-GPU scaling
    target is stated on), landmark-sharded over the ranks like the headline window.  Every rank builds the same seeded
    window and keeps its contiguous CSR range; timing = barrier + synchronize on both sides, max over ranks."""
    import torch
    from pvio_amd import BASummary, synth
    pb_full = synth.make_window(n_frames=n_frames, n_landmarks=n_landmarks, use_inertial=True, preintegrate=preintegrate)
    single = None
    if world > 1:  # the same window unsharded on rank 0's GPU, in this run
        if rank == 0:
            from pvio_amd.solver import HipContext
            c1 = HipContext(device=torch.cuda.current_device(), use_graph=not args.no_graph)
            c1.upload(pb_full)
            s1 = BASummary(pb_full, trace=False)
            for _ in range(max(warmup, 3)):
                c1.solve_resident(s1)
            torch.cuda.synchronize()
            t1, it1 = time.perf_counter(), 0
            for _ in range(steps):
                c1.solve_resident(s1)
                it1 += s1.num_iterations
            torch.cuda.synchronize()
            single = {"value": it1 / (time.perf_counter() - t1), "unit": "iterations/s", "steps": steps, "what": "the same window unsharded on rank 0's GPU, this run"}
            c1.close()
        dist.barrier()
    pb = pb_full.shard(rank, world)
    ctx.upload(pb)
    sm = BASummary(pb, trace=False)
    for _ in range(warmup):
        ctx.solve_resident(sm)
    barrier()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(steps):
        ctx.solve_resident(sm)
        iters += sm.num_iterations
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    prof = ctx.profile_resident(BASummary(pb, trace=False))
    out = {"workload": "%d KF x %d landmarks, full VIO factor set, %d reprojection factors (%d on this rank)" % (n_frames, n_landmarks, pb_full.n_obs, pb.n_obs),
           "value": iters / elapsed, "unit": "iterations/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "iterations_per_solve": iters / steps, "final_cost": float(sm.final_cost),
           "kernel_us_rank0": {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items()}}
    if world > 1 or args.force_sharded:  # the latency budget of an iteration, exchange steps included (eager launches, hipEvents)
        out["exchange_us_rank0"] = {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in ctx.last_comm.items()}
    if single:
        out["single_gpu_same_run"] = single
        out["speedup_vs_single_gpu"] = out["value"] / single["value"]
    return out


def relaunch_command(n, argv):
    """argv of the one-rank-per-GPU launch of this script (the form the driver uses for N > 1)"""
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", port,
            os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="vio", help="vio | vision | <N>x<M>_vio | <N>x<M>_vision")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline sample budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-klt", action="store_true")
    ap.add_argument("--no-scaling-window", action="store_true", help="skip the 10 KF x 50 000 landmark leg")
    ap.add_argument("--force-sharded", action="store_true",
                    help="diagnostics: run the multi-GPU code path (process group, RCCL communicator, eager launches, all-reduces) with "
                         "however many ranks there are, also one")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic: null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus 8` without a launcher used to run on ONE GPU and print "n_gpus": 8.  A bare multi-GPU
        # invocation now becomes the launch the contract describes (one rank per GPU, rendezvous on 127.0.0.1); the line's n_gpus is the
        # process group's world size, never the flag.
        os.execv(sys.executable, relaunch_command(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    # stdout carries exactly ONE line (the JSON): libraries that write to file descriptor 1 (RCCL prints its version banner
    # there at communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from pvio_amd import BAState, BASummary, capi, synth
    from pvio_amd.solver import HipContext, preintegrate

    lib = capi.load()
    # `value` is the window the metric is quoted on (BASELINE.json: 10 KF x 1000 landmarks) at EVERY N, so that the driver's N = 1, 2,
    # 4, 8 lines are one curve.  (Round 2 switched the headline to the 10 x 50 000 window for N > 1: lines of different workloads are not
    # comparable -- ADVICE r2.)  That window is latency-bound per iteration and cannot gain from landmark shards; the window
    # north_star states the multi-GPU target on -- 10 KF x 50 000 landmarks -- is reported by every line as `scaling_window`, with the
    # same window's unsharded rate on rank 0 measured in the same run (`single_gpu_same_run`) and the ratio of the two.
    pb_full, (n_frames, n_lm, vio) = build_window(args, preintegrate)
    single_gpu = None
    if world > 1:
        if rank == 0:
            c1 = HipContext(device=local_rank, use_graph=not args.no_graph)
            c1.upload(pb_full)
            s1 = BASummary(pb_full, trace=False)
            for _ in range(10):
                c1.solve_resident(s1)
            torch.cuda.synchronize()
            t1, it1, n1 = time.perf_counter(), 0, max(5, min(args.steps, 20))
            for _ in range(n1):
                c1.solve_resident(s1)
                it1 += s1.num_iterations
            torch.cuda.synchronize()
            single_gpu = {"value": it1 / (time.perf_counter() - t1), "unit": "iterations/s", "steps": n1, "what": "the same window unsharded on rank 0's GPU, this run"}
            c1.close()
        dist.barrier()
    pb = pb_full.shard(rank, world)
    ctx = HipContext(device=local_rank, rank=rank, world_size=world, use_graph=not args.no_graph, force_sharded=args.force_sharded)
    if sharded:
        import ctypes as C
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            assert lib.pvio_hip_comm_unique_id(uid) == 0
        t = torch.tensor(list(uid), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, 0)
        uid = (C.c_uint8 * 128)(*t.cpu().tolist())
        rc = lib.pvio_hip_comm_init(ctx.ctx, uid, rank, world)
        if rc != 0:
            raise SystemExit("pvio_hip_comm_init failed: %d" % rc)
    ctx.upload(pb)  # inputs resident in HBM before the timed region

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sm = BASummary(pb, trace=False)
    for _ in range(args.warmup):
        ctx.solve_resident(sm)
    barrier()
    t0 = time.perf_counter()
    iters = 0
    dev_s = 0.0
    for _ in range(args.steps):
        ctx.solve_resident(sm)  # returns after the stream has drained (ctrl block read back)
        iters += sm.num_iterations
        dev_s += sm.device_seconds
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- "API" rate: complete solves as the tracker issues them -- upload of the window (H2D), the iterations, the
    # states read back (D2H) -- next to the resident rate above ----
    api = None
    if rank == 0 and world == 1 and not args.force_sharded:
        n_api = max(10, min(args.steps, 100))
        # The C-ABI call is what is timed -- what a compiled caller (the BundleAdjustor adapter) pays.  The ctypes structs are built once and the
        # state arrays are reset outside the timed region: building them per call is ~70 us of Python, none of it the product's
        # (`ms_per_solve_python` = the same call through HipContext.solve(), marshalling included, which is what this leg reported until round 3).
        import ctypes as C
        pb_c, st_api, sm_api = pb.as_c(), BAState(pb), BASummary(pb, trace=False)
        st_c = st_api.as_c()
        init_fs, init_rho = st_api.frame_state.copy(), st_api.lm_inv_depth.copy()

        def api_call():
            np.copyto(st_api.frame_state, init_fs), np.copyto(st_api.lm_inv_depth, init_rho)
            t1 = time.perf_counter()
            rc = lib.pvio_hip_ba_solve(ctx.ctx, C.byref(pb_c), C.byref(st_c), C.byref(sm_api.c))
            dt = time.perf_counter() - t1
            if rc != 0:
                raise SystemExit("pvio_hip_ba_solve failed: %d" % rc)
            return dt

        for _ in range(3):
            api_call()
        api_iters, api_t = 0, []
        for _ in range(n_api):
            api_t.append(api_call())
            api_iters += sm_api.num_iterations
        py_t = []
        for _ in range(max(10, n_api // 4)):
            t1 = time.perf_counter()
            ctx.solve(pb, trace=False)
            py_t.append(time.perf_counter() - t1)
        med = float(np.median(api_t))  # the HIP runtime torch brings stalls for ~30 ms once in a few hundred calls: median, mean beside it
        api = {"value": (api_iters / n_api) / med, "unit": "iterations/s", "ms_per_solve": 1e3 * med, "ms_per_solve_mean": 1e3 * sum(api_t) / n_api,
               "ms_per_solve_python": 1e3 * float(np.median(py_t)), "solves": n_api,
               "what": "the pvio_hip_ba_solve call itself: staging + upload (one DMA) + iterations + read-back of the states, same window every time; median over the solves"}
        ctx.upload(pb)  # back to the resident state the legs below expect

    # ---- pvio_hip_opts::reuse_identical_candidates (NOT the headline: `value` above evaluates every candidate, like the reference) ----
    # The same resident solves with the short-circuit on: a candidate that is bit-identical to the one just rejected is not evaluated again.
    # Same iterations, records and results; the window's last four iterations re-evaluate one rejected point (|gn| << radius), three of the
    # evaluations are saved.  Reported because it is what the host adapter runs by default -- the reference's keyframe solves consist of such runs.
    reuse = None
    if rank == 0 and world == 1 and not args.force_sharded:
        try:
            c2 = HipContext(device=local_rank, use_graph=not args.no_graph, reuse_identical_candidates=True)
            c2.upload(pb)
            sm2 = BASummary(pb, trace=False)
            for _ in range(3):
                c2.solve_resident(sm2)
            n2 = max(5, min(args.steps, 100))
            torch.cuda.synchronize()
            t1, it2 = time.perf_counter(), 0
            for _ in range(n2):
                c2.solve_resident(sm2)
                it2 += sm2.num_iterations
            dt2 = time.perf_counter() - t1
            reuse = {"value": it2 / dt2, "unit": "iterations/s", "ms_per_step": 1e3 * dt2 / n2, "steps": n2, "iterations_per_solve": it2 / n2,
                     "candidate_evaluations_saved_per_solve": c2.last_candidate_repeats(), "final_cost": sm2.final_cost,
                     "what": "pvio_hip_opts::reuse_identical_candidates = 1: bit-identical re-evaluations of a rejected candidate skipped; results identical; not the headline"}
            c2.close()
        except Exception as e:
            reuse = {"error": repr(e)}

    # ---- roofline leg: per-kernel durations from hipEvents on the solver's stream ----
    prof = ctx.profile_resident(BASummary(pb, trace=False))
    prof = ctx.profile_resident(BASummary(pb, trace=False))
    alg_bytes = synth.algorithmic_bytes_per_iteration(pb)
    lin_ms, lin_n = prof["k_linearize"]
    avg_s = (lin_ms / max(lin_n, 1)) * 1e-3
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    roofline = {
        "bound": "hbm", "kernel": "k_linearize", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("k_linearize", args),
        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": avg_s * 1e6,
        "kernel_us": {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items()},
        "kernel_us_what": "hipEvents on the solver's stream around EAGER launches (one slot at a time, host synchronization per slot): their sum exceeds "
                          "device_ms_per_step / slots, which is a hipGraph replay; kernel_us_rocprof = the same kernels inside graph replays",
        "exchange_us": {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in ctx.last_comm.items()} if sharded else None,
        "note": "working set < 1 MB: L2/Infinity-Cache resident, the iteration is launch/dependency-latency bound",
    }
    # the roofline above is for the kernel that moves the window's data (SURVEY 8d's per-iteration bytes); by DURATION the
    # longest launch of a small window is the one-workgroup dense solve, a serial FP64 chain with no bandwidth or matrix-core
    # roofline worth quoting (P^3/3 flops in its time is < 0.1 % of the FP64 peak)
    roofline["longest_kernel"] = max(roofline["kernel_us"], key=roofline["kernel_us"].get)
    roofline["traffic_source"] = _PMC["note"]
    if _PMC.get("stats"):  # rocprofv3 --kernel-trace --stats child pass of this run (graph replays): what profiles/*kernel_stats*.csv holds
        roofline["kernel_us_rocprof"] = {k: v["avg_us"] for k, v in _PMC["stats"].items() if k in roofline["kernel_us"]}
        lin_r = _PMC["stats"].get("k_linearize")
        if lin_r:
            roofline["achieved_rocprof"] = alg_bytes / (lin_r["avg_us"] * 1e-6) / 1e9
            roofline["frac_rocprof"] = roofline["achieved_rocprof"] / HBM_PEAK_GBS
    # the dense kernel against the only roofline it has, FP64 arithmetic: P^3 / 3 + 2 P^2 flops (Cholesky of the reduced system with the
    # right-hand side carried along + back substitution) per FACTORING launch.  Its average duration over all launches of a solve
    # (rejected steps do not factor) is a lower bound of a factoring launch's: the fraction quoted is an upper bound -- and still
    # ~1e-4: one workgroup on one CU, a chain of P dependent pivots (profiles/NOTES_r1_r3.md section 4).
    Pd = (15 if vio else 6) * n_frames
    dense_flops = Pd ** 3 / 3.0 + 2.0 * Pd ** 2
    dense_s = roofline["kernel_us"]["k_dense"] * 1e-6
    roofline_dense = {"bound": "mfma", "kernel": "k_dense", "achieved": dense_flops / dense_s / 1e12 if dense_s > 0 else 0.0, "peak": FP64_PEAK_TFLOPS,
                      "unit": "TFLOP/s", "frac": (dense_flops / dense_s / 1e12) / FP64_PEAK_TFLOPS if dense_s > 0 else 0.0,
                      "traffic": pmc_traffic("k_dense", args), "flops_per_factoring_launch": dense_flops, "avg_launch_us": dense_s * 1e6,
                      "note": "dtype f64 on the FP64 matrix cores (v_mfma_f64_16x16x4_f64) + FP64 VALU; one workgroup: a serial dependency chain, not a throughput kernel"}
    if n_lm >= 10000:
        roofline["note"] = ("large window: k_linearize is FP64-issue bound (factor evaluation on the VALU + Schur complement on the f64 matrix "
                            "cores), not HBM bound; see profiles/NOTES_r1_r3.md section 5")

    # ---- scaling window: the window north_star's multi-GPU sentence names (10 KF x 50 000 landmarks, full factor set),
    # sharded exactly like the headline window, same context and communicator; a few solves, barrier-bracketed ----
    scaling_window = None
    if not args.no_scaling_window and args.workload in ("vio", "vision"):
        try:
            scaling_window = bench_scaling_window(ctx, args, rank, world, dist, barrier, preintegrate)
        except Exception as e:  # the headline line must still be printed
            scaling_window = {"error": repr(e)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as O
        O.build()

        def time_cpu(solve, budget):
            st, so = BAState(pb_full), BASummary(pb_full, trace=False)
            solve(pb_full, st, so)  # warm-up
            c_it, c_t, c_n = 0, 0.0, 0
            while c_t < budget and c_n < 2000:
                st, so = BAState(pb_full), BASummary(pb_full, trace=False)
                t1 = time.perf_counter()
                solve(pb_full, st, so)
                c_t += time.perf_counter() - t1
                c_it += so.num_iterations
                c_n += 1
            return c_it / c_t, c_n, c_t
        # two builds of the same restatement are timed: release style (-O3 -march=native of this host, FMA contraction on) and
        # the checker build the parity tests use (-ffp-contract=off); the FASTER one is quoted (contraction does not always pay)
        v_fast, c_n, c_t = time_cpu(O.solve_fast, 0.7 * args.cpu_seconds)
        v_chk, _, _ = time_cpu(O.solve, 0.3 * args.cpu_seconds)
        cpu = {"value": max(v_fast, v_chk), "unit": "BA iterations/s", "cores": 1, "kind": "port",
               "sample": "%d solves of the same window (%.1f s), single thread as the reference's num_threads=1 "
                         "(solver_options.h:31); host has %d cores" % (c_n, c_t, os.cpu_count()),
               "build": "g++ -O3 -march=native (compiled on this host), FMA contraction on; dense-Schur restatement of the Ceres path "
                        "(the reference configures SPARSE_SCHUR; it cannot be built here)",
               "value_fast_build": v_fast, "value_checker_build": v_chk, "quoted": "the faster of the two builds", "checker_build": "-O3 -march=native -ffp-contract=off (the build the parity tests compare against)"}

    # ---- KLT leg (second half of BASELINE.json's metric): tracks/ms, pyramids resident, TUM-VI-sized frames ----
    klt = None
    if rank == 0 and world == 1 and not args.no_klt:
        klt = bench_klt(ctx, args)

    multi = None
    if rank == 0 and world == 1 and not args.no_klt:
        multi = bench_concurrent_windows(pb_full)

    ref_window = None
    if rank == 0 and world == 1 and not args.no_klt and args.workload == "vio":
        try:
            ref_window = bench_reference_window(preintegrate, cpu_seconds=0.0 if args.no_cpu_baseline else 2.0)
        except Exception as e:  # the headline line must still be printed
            ref_window = {"error": repr(e)}

    if rank == 0:
        value = iters / elapsed
        out = {
            "metric": "BA iterations/sec", "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d KF x %d landmarks, %s, %d reprojection factors, <=%d trust-region iterations per solve"
                                   % (n_frames, n_lm, "full VIO factor set (IMU pre-integration + gauge prior)" if vio else "reprojection only",
                                      pb_full.n_obs, pb_full.max_iterations),
                       "parallelism": "landmark shards x%d, RCCL all-reduce of the reduced pose system" % world if sharded else "single GPU",
                       "graph": not args.no_graph},  # sharded solves are graph-captured too (collectives included) unless PVIO_HIP_SHARDED_GRAPH=0
            "iterations_per_solve": iters / args.steps,
            "device_ms_per_step": 1e3 * dev_s / args.steps,
            "api": api,
            "identical_candidate_reuse": reuse,
            "final_cost": sm.final_cost,
            "roofline": roofline,
            "roofline_dense": roofline_dense,
            "cpu_baseline": cpu,
            "klt": klt,
            "concurrent_windows": multi,
            "scaling_window": scaling_window,
            "reference_window": ref_window,
        }
        if world > 1:
            out["single_gpu_same_workload"] = single_gpu
            if single_gpu:
                out["speedup_vs_single_gpu"] = value / single_gpu["value"]
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
