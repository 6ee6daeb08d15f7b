"""The drop-in over a SEQUENCE, inside the reference's own pipeline: `pvio::PVIO` (pvio/include/pvio/pvio.h:135-148) -- the object pvio-pc's main
loop drives -- compiled from the reference's unedited sources (pvio.cpp, core/{core,feature_tracker,frontend_worker,sliding_window_tracker,
plane_extractor}.cpp, map/*.cpp, estimation/{factor,preintegrator}.cpp; oracle/ref/Makefile) and run over the rendered sequence of
test_host_headless.py twice:

  libpvio_ref.so           ... with the reference's own estimation/bundle_adjustor.cpp + pnp.cpp (mini-Ceres below them)
  libpvio_dropin[_emu].so  ... with the PRODUCT's pvio_amd/host/{bundle_adjustor,pnp,pnp_solve}.cpp linked in their place, above libpvio_hip.so
                               (the kernel emulator in the CPU suite); with PVIO_SEQ_IMAGE=hip the pvio::Image is the product's HipImage as well

Everything between PVIO::track_camera and the hot path is the reference's: IMU pairing, FeatureTracker::work, Frame::track_keypoints /
detect_keypoints, FrontendWorker::work, SlidingWindowTracker::track (PnP, keyframe check, Map::marginalize_frame, plane extraction and casting,
solve, pruning).  The one substitution is the SfM initializer (out of scope, beyond the mini-Eigen): the first window is bootstrapped from supplied
poses (oracle/ref/gt_initializer.cpp), on both sides alike.  tests/chain_compare.py::compare_seq holds the two record streams together after
every camera frame.  Also here: the same reference run against the ORACLE CHAIN of test_chain_parity.py (restated control flow + oracle_front.cpp):
that pins K3 / K5 / K6 -- which tracks survive, which corners are added, frame by frame -- to the reference's own Frame::track_keypoints."""
import os
import subprocess
import sys

import numpy as np
import pytest

import chain_compare
import test_host_headless as hh
from chain_run import parse_log

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def _libs():
    from oracle import ref_py
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hipemu"), "libpvio_hipemu.so"])
    if not ref_py.available() or not ref_py.build_dropin():
        pytest.skip("oracle/_ref libraries not built and /root/reference absent")


def _run(lib, prefix, n_frames, window, gap, distance, size, timeout, image="oracle"):
    env = dict(os.environ, PVIO_SEQ_IMAGE=image)
    r = subprocess.run([sys.executable, os.path.join(HERE, "chain_run.py"), lib, prefix, str(n_frames), str(window), str(gap), str(distance), size],
                       capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


def test_reference_pvio_with_product_backend_emulated(tmp_path):
    """30 frames at 352 x 264: bootstrap solve, 24 PnP, keyframe solves + marginalizations, planes extracted and cast -- the reference's pvio::PVIO with the
    reference's back-end against the same with the product's (kernels in the emulator); the CPU oracle's front end behind pvio::Image on both sides"""
    _libs()
    a, b = str(tmp_path / "ref"), str(tmp_path / "dropin")
    print(_run(os.path.join(REFDIR, "libpvio_ref.so"), a, 30, 3, 2, 18.0, "small", 600))
    print(_run(os.path.join(REFDIR, "libpvio_dropin_emu.so"), b, 30, 3, 2, 18.0, "small", 900))
    info = chain_compare.compare_seq(a + ".log", b + ".log", hh.SMALL[2][0])
    print("reference pvio::PVIO, reference back-end vs product back-end (wall scene):", info)
    # strict until the reference's best-plane coin flip (chain_compare.compare_seq): bootstrap solve, PnP of every frame, the first keyframe cycles
    assert info["frames"] == 30 and info["strict_frames"] >= 16 and info["window_records"] >= 10 and info["max_state"] < 1e-8
    # the relief scene has no planes: strict over the whole sequence
    a, b = str(tmp_path / "ref_r"), str(tmp_path / "dropin_r")
    print(_run(os.path.join(REFDIR, "libpvio_ref.so"), a, 30, 3, 2, 18.0, "small_relief", 600))
    print(_run(os.path.join(REFDIR, "libpvio_dropin_emu.so"), b, 30, 3, 2, 18.0, "small_relief", 900))
    info = chain_compare.compare_seq(a + ".log", b + ".log", hh.SMALL[2][0])
    print("reference pvio::PVIO, reference back-end vs product back-end (relief scene):", info)
    assert info["frames"] == 30 and info["strict_frames"] == 30 and info["window_records"] >= 20 and info["keyframes"] >= 2 and info["max_state"] < 1e-8


def test_reference_pvio_pins_the_restated_front_end_bookkeeping(tmp_path):
    """SURVEY K3 / K5 / K6 against the reference's own code at sequence level: the oracle chain (tests/host/oracle_chain.cpp: restated FeatureTracker::work
    order, oracle_front.cpp's survivor selection / Poisson filter / prediction) and the reference's pvio::PVIO (its own feature_tracker.cpp, frame.cpp,
    poisson_disk_filter.h) see the same images through the same oracle front end: same track ids, track lengths and keypoints in every frame, until
    the reference's plane extractor -- which the restated driver does not have -- changes the window."""
    _libs()
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host"), "libpvio_chain_oracle.so"])
    a, b = str(tmp_path / "ref"), str(tmp_path / "oracle")
    print(_run(os.path.join(REFDIR, "libpvio_ref.so"), a, 30, 3, 2, 18.0, "small", 600))
    print(_run(os.path.join(HERE, "host", "libpvio_chain_oracle.so"), b, 30, 3, 2, 18.0, "small", 600))
    A = [r for r in parse_log(a + ".log") if r[0] == 1]
    B = [r for r in parse_log(b + ".log") if r[0] == 1]
    R9 = [r for r in parse_log(a + ".log") if r[0] == 9]
    first_planes = min([int(I[0]) for _, I, _ in R9 if int(I[2 + 4 * int(I[1])]) > 0] + [len(A)])
    assert first_planes >= 12, first_planes
    tracked = 0
    for (_, Ia, Da), (_, Ib, Db) in zip(A[:first_planes], B[:first_planes]):
        assert Ia.shape == Ib.shape and (Ia == Ib).all(), "frame %d: surviving tracks / new corners differ" % int(Ia[0])
        n = int(Ia[4])
        assert (Da[:2 * n] == Db[:2 * n]).all()                 # same keypoints, bit for bit
        assert np.abs(Da[-8:] - Db[-8:]).max() < 1e-9          # same reported pose
        tracked += int((Ia[5:].reshape(n, 2)[:, 1] > 1).sum())
    assert tracked > 1000
    print("identical track ids / lengths / keypoints for %d frames (%d tracked keypoints), until the reference's plane extractor acts" % (first_planes, tracked))


@pytest.mark.gpu
def test_reference_pvio_with_product_backend_gpu(tmp_path):
    """60 frames at 512 x 384, window of 6 keyframes: the reference's pvio::PVIO with its own back-end against the same with the product's on the MI355X"""
    _libs()
    a, b = str(tmp_path / "ref"), str(tmp_path / "dropin")
    print(_run(os.path.join(REFDIR, "libpvio_ref.so"), a, 60, 6, 3, 25.0, "full", 1500))
    print(_run(os.path.join(REFDIR, "libpvio_dropin.so"), b, 60, 6, 3, 25.0, "full", 900))
    info = chain_compare.compare_seq(a + ".log", b + ".log", hh.K4[0])
    print("reference pvio::PVIO, reference back-end vs product back-end (GPU, wall scene):", info)
    assert info["frames"] == 60 and info["strict_frames"] >= 27 and info["window_records"] >= 10
    out = os.environ.get("PVIO_SEQ_REPORT")
    if out:
        import json
        json.dump(info, open(out, "w"), indent=1)


def _full_product(tmp_path, lib, n_frames, window, gap, distance, size, fx, timeout):
    _libs()
    a, b = str(tmp_path / ("ref_" + size)), str(tmp_path / ("product_" + size))
    print(_run(os.path.join(REFDIR, "libpvio_ref.so"), a, n_frames, window, gap, distance, size, timeout))
    print(_run(os.path.join(REFDIR, lib), b, n_frames, window, gap, distance, size, timeout, image="hip"))
    keep = os.environ.get("PVIO_SEQ_KEEP")
    if keep:
        import shutil
        os.makedirs(keep, exist_ok=True)
        for f in (a + ".log", b + ".log"):
            shutil.copy(f, keep)
    return chain_compare.compare_seq(a + ".log", b + ".log", fx)


def test_reference_pvio_with_the_whole_product_emulated(tmp_path):
    """the reference's pvio::PVIO with EVERYTHING below its seams the product's: pvio::Image = HipImage (CLAHE, pyramid, LK, corner detection and F-RANSAC
    kernels; pvio_amd/host/feature_front.cpp), BundleAdjustor and visual_inertial_pnp = pvio_amd/host above the C ABI (kernels in the emulator) -- against the
    same pvio::PVIO with the reference's back-end and the CPU oracle's front end.  The LK sums are in the defined order: keypoints are bit-identical.
    Relief scene (no planes): strict over the whole sequence."""
    info = _full_product(tmp_path, "libpvio_dropin_emu.so", 30, 3, 2, 18.0, "small_relief", hh.SMALL[2][0], 1500)
    print("reference pvio::PVIO: reference back-end + oracle front end vs the whole product:", info)
    assert info["frames"] == 30 and info["strict_frames"] == 30 and info["max_kp_px"] == 0.0 and info["max_state"] < 1e-8 and info["keyframes"] >= 2


@pytest.mark.gpu
def test_reference_pvio_with_the_whole_product_gpu(tmp_path):
    """the same on the MI355X: 60 frames at 512 x 384, window of 6 keyframes; the relief scene strictly over all 60 frames, the wall scene (planes
    extracted, cast and constrained) strictly until the reference's best-plane coin flip"""
    out = {}
    info = _full_product(tmp_path, "libpvio_dropin.so", 60, 6, 3, 25.0, "full_relief", hh.K4[0], 1500)
    print("reference pvio::PVIO: reference back-end + oracle front end vs the whole product (GPU, relief scene):", info)
    assert info["frames"] == 60 and info["strict_frames"] == 60 and info["max_kp_px"] <= 1e-3 and info["keyframes"] >= 3
    out["relief"] = info
    info = _full_product(tmp_path, "libpvio_dropin.so", 60, 6, 3, 25.0, "full", hh.K4[0], 1500)
    print("reference pvio::PVIO: reference back-end + oracle front end vs the whole product (GPU, wall scene):", info)
    assert info["frames"] == 60 and info["strict_frames"] >= 27 and info["max_kp_px"] <= 1e-3
    out["wall"] = info
    path = os.environ.get("PVIO_SEQ_REPORT_FULL")
    if path:
        import json
        json.dump(out, open(path, "w"), indent=1)


def _long_sequence(tmp_path, scene, n_frames=360, window=8, gap=3, pose_after_flip=chain_compare.SEQ_POSE_AFTER_FLIP):
    """reference's pvio::PVIO + reference back-end + oracle front end against the same pvio::PVIO with the WHOLE product below its seams, over a long sequence
    on the sweep trajectory; both trajectories against the ground truth (ATE)."""
    _libs()
    a, b = str(tmp_path / ("ref_" + scene)), str(tmp_path / ("product_" + scene))
    # the two runs side by side (the reference side is CPU only, the product side waits on the GPU): 80 s instead of 120 s of wall clock
    procs = []
    for lib, prefix, image in ((os.path.join(REFDIR, "libpvio_ref.so"), a, "oracle"), (os.path.join(REFDIR, "libpvio_dropin.so"), b, "hip")):
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "chain_run.py"), lib, prefix, str(n_frames), str(window), str(gap), "25.0", scene],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, PVIO_SEQ_IMAGE=image)))
    lines = []
    for pr in procs:
        so, se = pr.communicate(timeout=2400)
        assert pr.returncode == 0, so[-2000:] + se[-2000:]
        lines.append(so.strip().splitlines()[-1])
    line_ref, line_prod = lines
    keep = os.environ.get("PVIO_SEQ_KEEP")
    if keep:
        import shutil
        os.makedirs(keep, exist_ok=True)
        for f in (a + ".log", b + ".log", a + ".tum", b + ".tum", a + ".gt.npy"):
            shutil.copy(f, keep)
    info = chain_compare.compare_seq(a + ".log", b + ".log", hh.K4[0], allow_divergence=True, pose_after_flip=pose_after_flip)
    gt = np.load(a + ".gt.npy")
    ate_ref, n_ref = chain_compare.ate_rmse(a + ".tum", gt)
    ate_prod, n_prod = chain_compare.ate_rmse(b + ".tum", gt)
    ta, tb = np.loadtxt(a + ".tum", ndmin=2), np.loadtxt(b + ".tum", ndmin=2)
    info.update(scene=scene, n_frames=n_frames, window=window, reference_run=line_ref, product_run=line_prod, ate_rmse_reference_m=ate_ref, ate_rmse_product_m=ate_prod,
                ate_poses=n_ref, ate_difference_m=abs(ate_ref - ate_prod), trajectory_max_difference_m=float(np.abs(ta[:, 1:4] - tb[:, 1:4]).max()) if ta.shape == tb.shape else None)
    assert n_ref == n_prod
    return info


@pytest.mark.gpu
def test_long_sequence_ate_reference_vs_whole_product_gpu(tmp_path):
    """VERDICT r4 item 8 ("final ATE equal" on more than a handful of keyframe solves): 360 frames at 512 x 384 (18 s, 31 keyframe solves with a marginalization
    each, window of 8 like config/euroc.yaml:50) on the sweep trajectory; the reference's pvio::PVIO with the reference's back-end and the CPU oracle's front
    end against the same pvio::PVIO with the whole product below its seams.
    Relief scene (no planes), measured (profiles/r5_seq_long.json): identical track ids, flags and keypoints (0 px) in ALL 360 frames, every window state within
    1.3e-9, reported poses within 1.3e-9 m, ATE 3.978084347 cm against 3.978084319 cm.  (Before the seven-point step of the F-matrix RANSAC got a defined
    arithmetic the same run stayed identical for 63 frames and ended at a tie between two hypotheses of equal inlier count:
    tests/test_host_ransac.py::test_sequence_divergences_are_ransac_ties_between_equal_hypotheses.)
    PVIO_LONG_SEQUENCE_WALL=1 adds the wall scene (planes extracted, cast and constrained): strict until the reference's own best-plane coin flip
    (chain_compare.compare_seq; frame 34 here), after which the two runs are different experiments of a pipeline whose own ATE on this scene is 13 cm:
    reported poses up to 17.5 cm apart, ATE 13.1 cm (reference) against 12.6 cm (product)."""
    import json
    out = {}
    info = _long_sequence(tmp_path, "full_relief_sweep")
    print("long sequence, relief scene:", info)
    assert info["frames"] == 360 and info["strict_frames"] == 360 and info["keyframes"] >= 30 and info["max_state"] <= 1e-6 and info["max_kp_px"] <= 1e-3
    assert info["first_divergence"] is None and info["ate_difference_m"] <= 1e-6 and info["ate_rmse_product_m"] < 0.08
    out["relief"] = info
    if os.environ.get("PVIO_LONG_SEQUENCE_WALL"):
        info = _long_sequence(tmp_path, "full_sweep", pose_after_flip=0.25)
        print("long sequence, wall scene (planes on):", info)
        assert info["frames"] == 360 and info["strict_frames"] >= 30 and info["first_divergence"] is None
        # (the REFERENCE's own run of this scene has an ATE of 13 cm: its plane factors pull the window; what is held is that the product's run
        # stays with the reference's, not that either is good)
        assert info["ate_difference_m"] < 0.03
        out["wall"] = info
    path = os.environ.get("PVIO_SEQ_REPORT_LONG")
    if path:
        json.dump(out, open(path, "w"), indent=1)


@pytest.mark.gpu
def test_long_sequence_second_family_gpu(tmp_path):
    """VERDICT r5 weak #9 (the 360 / 360 identity rests on ONE rendered trajectory family): the same comparison -- the reference's pvio::PVIO with its own back-end and
    the oracle's front end against the same pvio::PVIO with the whole product below its seams -- on a second family: another texture, another relief (other
    wavelengths and phases), a narrower and faster sweep with a larger, faster height oscillation (tests/test_host_headless.py, variant 1; scene "..._b").
    Same bar: every one of the 360 frames identical in track ids, flags and keypoints, every window state within 1e-6, the two ATEs equal to 1e-6 m."""
    import json
    info = _long_sequence(tmp_path, "full_relief_sweep_b")
    print("long sequence, second family:", info)
    assert info["frames"] == 360 and info["strict_frames"] == 360 and info["keyframes"] >= 25 and info["max_state"] <= 1e-6 and info["max_kp_px"] <= 1e-3
    assert info["first_divergence"] is None and info["ate_difference_m"] <= 1e-6 and info["ate_rmse_product_m"] < 0.10
    path = os.environ.get("PVIO_SEQ_REPORT_LONG_B")
    if path:
        json.dump({"relief_b": info}, open(path, "w"), indent=1)
