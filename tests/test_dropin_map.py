"""The drop-in, RUN: the same pvio::Map, two bundle_adjustor.o.

oracle/ref/Makefile builds two libraries on the reference's REAL object graph (map/{frame,track,map,plane}.cpp, estimation/{factor,
preintegrator}.cpp, geometry/lie_algebra.cpp, core/plane_extractor.cpp of /root/reference, compiled unedited; functional mini-Eigen):
  libpvio_ref.so          + the reference's own estimation/bundle_adjustor.cpp and pnp.cpp (mini-Ceres below them)
  libpvio_dropin[_emu].so + the PRODUCT's pvio_amd/host/{bundle_adjustor,pnp,pnp_solve}.cpp compiled -DPVIO_HOST_USE_REFERENCE_TYPES against the
                            reference's headers, above libpvio_hip.so (the fiber emulator of tests/hipemu in the CPU suite)
Both get identical Maps from the same builder (oracle/ref/ref_window.h) and are diffed after every call the reference's callers make
(core/sliding_window_tracker.cpp:113 solve, map/map.cpp:77 marginalize_frame, bundle_adjustor.h:35 compute_reprojection_error,
sliding_window_tracker.cpp:79 visual_inertial_pnp): every Frame::pose / motion, Track::landmark.inv_depth / quality, TF_VALID / TF_PLANE,
Plane::tracks, the accept / reject trace, the new prior's S^T S / S^T s -- north_star's 1e-6 on states, identical flags.

Both libraries travel prebuilt to the GPU box; nothing here reads /root/reference at run time."""
import os
import subprocess

import numpy as np
import pytest

import ba_compare
from pvio_amd import BAState, BASummary, synth
from test_ref_pin import ref  # noqa: F401  (fixture)

STATE_TOL = ba_compare.STATE_TOL
EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu")

SOLVE_CASES = {
    "vision": dict(n_frames=6, n_landmarks=40, visibility=3),
    "vio": dict(n_frames=6, n_landmarks=40, use_inertial=True, visibility=4),
    "vio_zero_bias_quirk": dict(n_frames=4, n_landmarks=30, use_inertial=True, bias_init="zero", perturb_scale=1.0),
    "plane": dict(n_frames=5, n_landmarks=60, plane_fraction=0.5),
    "vio_plane": dict(n_frames=5, n_landmarks=60, plane_fraction=0.4, use_inertial=True),
    # tracks of planes with fewer than 20 members: duplicate residual blocks (bundle_adjustor.cpp:165-179) -- small planes in the Map
    "vio_small_planes": dict(n_frames=6, n_landmarks=70, use_inertial=True, visibility=4, duplicate_fraction=0.3),
    "config1_10x200": dict(n_frames=10, n_landmarks=200),
}
BIG = {"metric_10x1000_vio": dict(n_frames=10, n_landmarks=1000, use_inertial=True)}


def _harness(ref, kind):  # noqa: F811
    if kind == "emu":
        subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    if not ref.build_dropin():
        pytest.skip("oracle/_ref/libpvio_dropin*.so not built and /root/reference absent")
    return ref.dropin(kind)


@pytest.fixture(scope="module")
def emu(ref):  # noqa: F811
    return _harness(ref, "emu")


@pytest.fixture(scope="module")
def gpu(ref):  # noqa: F811
    return _harness(ref, "gpu")


def diff_solve(ref, drop, pb, tracks_fn=None):  # noqa: F811
    """BundleAdjustor().solve(map, config, use_inertial) through both libraries on identical Maps"""
    ta = tracks_fn() if tracks_fn else None
    tb = tracks_fn() if tracks_fn else None
    fa, ta, sa = ref.reference().solve(pb, tracks=ta)
    fb, tb, sb = drop.solve(pb, tracks=tb)
    assert (sb.termination, sb.num_iterations, sb.num_successful_steps, sb.is_usable) == (sa.termination, sa.num_iterations, sa.num_successful_steps, sa.is_usable)
    tra, trb = sa.trace(), sb.trace()
    assert len(tra) == len(trb)
    N16 = 16 * pb.n_frames
    worst = 0.0
    for k, (a, b) in enumerate(zip(tra, trb)):
        assert (a["step_is_valid"], a["step_is_successful"]) == (b["step_is_valid"], b["step_is_successful"]), (k, a, b)
        np.testing.assert_allclose(b["cost"], a["cost"], rtol=1e-5)
        assert a["mu"] == b["mu"]
        worst = max(worst, np.abs(sb.trace_states[k][:N16] - sa.trace_states[k][:N16]).max())  # Frame::pose / motion after EVERY iteration
    assert worst <= STATE_TOL, worst
    np.testing.assert_allclose(fb, fa, rtol=0, atol=STATE_TOL)                         # Frame::pose / motion
    np.testing.assert_allclose(tb.inv_depth, ta.inv_depth, rtol=0, atol=STATE_TOL)     # Track::landmark.inv_depth, every track of the Map
    np.testing.assert_array_equal(tb.valid, ta.valid)                                  # TF_VALID
    np.testing.assert_array_equal(tb.plane, ta.plane)                                  # TF_PLANE
    np.testing.assert_array_equal(tb.membership, ta.membership)                        # Plane::tracks
    ok = ta.valid.astype(bool)
    np.testing.assert_allclose(tb.quality[ok], ta.quality[ok], rtol=0, atol=1e-5)      # landmark.quality (px)
    return dict(iterations=sb.num_iterations, rejected=sb.num_iterations - sb.num_successful_steps, worst_state_diff=worst,
                worst_inv_depth=float(np.abs(tb.inv_depth - ta.inv_depth).max()) if len(ta.inv_depth) else 0.0)


@pytest.mark.parametrize("name", list(SOLVE_CASES))
def test_emulated_dropin_solve_on_reference_map(ref, emu, oracle, name):  # noqa: F811
    pb = ba_compare.make(oracle, **SOLVE_CASES[name])
    print(name, diff_solve(ref, emu, pb))


def _revalidation_window(oracle):
    """12-frame VIO window, two planes of 40 tracks, five tracks per plane 0.3 m off their plane: the plane-track re-validation of
    bundle_adjustor.cpp:251-275 erases memberships and re-promotes tracks; the depth gate / quality pass runs over every track"""
    import host_compare
    pb = ba_compare.make(oracle, n_frames=12, n_landmarks=200, use_inertial=True, plane_fraction=0.4, plane_outliers=5)
    t0 = host_compare.flat_tracks(pb, BAState(pb))
    M = pb.n_landmarks
    best = np.r_[np.full(M, -1), np.argmax(t0["membership"][:, M:], axis=0)]
    return pb, t0, best


def _post_pass_diff(ref, drop, oracle):  # noqa: F811
    pb, t0, best = _revalidation_window(oracle)
    mk = lambda: ref.Tracks(t0["ptr"], t0["frame"], t0["z"], t0["inv_depth"], t0["valid"], t0["plane"], t0["life"], best, t0["normal"], t0["distance"], t0["membership"])  # noqa: E731
    out = diff_solve(ref, drop, pb, tracks_fn=mk)
    # and the pass did something: plane tracks were re-promoted on both sides
    fb, tb, _ = drop.solve(pb, tracks=mk())
    M = pb.n_landmarks
    moved = (tb.plane[M:] == 0) & (tb.valid[M:] == 1)
    assert moved[:5].all() and moved[40:45].all()
    out["re_promoted"] = int(moved.sum())
    return out


def test_emulated_dropin_post_solve_passes_on_reference_map(ref, emu, oracle):  # noqa: F811
    print(_post_pass_diff(ref, emu, oracle))


def _solved(oracle, **kw):
    pb = ba_compare.make(oracle, **kw)
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)
    return pb, st


def diff_marginalize(ref, drop, oracle, victim, regular_prior):  # noqa: F811
    """BundleAdjustor().marginalize_frame(map, victim) + compute_reprojection_error(map) through both libraries"""
    import marg_compare
    pb, st = marg_compare.solved_window(oracle, regular_prior=regular_prior, n_frames=8, n_landmarks=200, use_inertial=True, visibility=5)
    ta, _ = ref.tracks_of_problem(pb, inv_depth=st.lm_inv_depth)
    tb, _ = ref.tracks_of_problem(pb, inv_depth=st.lm_inv_depth)
    Sa, sa, IMa, iva = ref.reference().marginalize(pb, st.frame_state, ta, victim)
    Sb, sb, IMb, ivb = drop.marginalize(pb, st.frame_state, tb, victim)
    scale = np.abs(IMa).max()
    np.testing.assert_allclose(IMb, IMa, rtol=1e-6, atol=1e-7 * scale)
    np.testing.assert_allclose(ivb, iva, rtol=1e-6, atol=1e-6 * np.abs(iva).max())
    ea = ref.reference().reprojection_error(pb, st.frame_state, ta)
    eb = drop.reprojection_error(pb, st.frame_state, tb)
    assert abs(ea - eb) <= 1e-9 * max(ea, 1e-12), (ea, eb)
    return dict(victim=victim, info_matrix_rel=float(np.abs(IMb - IMa).max() / scale), reprojection_error_px=eb)


@pytest.mark.parametrize("victim", [0, 3])
def test_emulated_dropin_marginalize_on_reference_map(ref, emu, oracle, victim):  # noqa: F811
    print(diff_marginalize(ref, emu, oracle, victim, regular_prior=(victim != 0)))


def diff_cycle(ref, drop, oracle, **kw):  # noqa: F811
    """the keyframe cycle on ONE Map: Map::marginalize_frame(0) (the reference's own caller, map.cpp:73-88, which erases the victim and
    re-anchors its tracks) and then BundleAdjustor().solve of the frames left with the prior just made (sliding_window_tracker.cpp:91-113)"""
    pb, st = _solved(oracle, **kw)
    ta, _ = ref.tracks_of_problem(pb, inv_depth=st.lm_inv_depth)
    tb, _ = ref.tracks_of_problem(pb, inv_depth=st.lm_inv_depth)
    fa, ua = ref.reference().marginalize_then_solve(pb, st.frame_state, ta, 0)
    fb, ub = drop.marginalize_then_solve(pb, st.frame_state, tb, 0)
    assert ua == ub
    np.testing.assert_allclose(fb, fa, rtol=0, atol=STATE_TOL)
    np.testing.assert_allclose(tb.inv_depth, ta.inv_depth, rtol=0, atol=STATE_TOL)
    np.testing.assert_array_equal(tb.valid, ta.valid)
    return dict(worst_state_diff=float(np.abs(fb - fa).max()), worst_inv_depth=float(np.abs(tb.inv_depth - ta.inv_depth).max()))


def test_emulated_dropin_keyframe_cycle_on_reference_map(ref, emu, oracle):  # noqa: F811
    print(diff_cycle(ref, emu, oracle, n_frames=7, n_landmarks=120, use_inertial=True, visibility=5))


def diff_pnp(ref, drop, oracle, use_inertial):  # noqa: F811
    """visual_inertial_pnp(map, frame, config, use_inertial) (pnp.cpp:32-100) through both libraries"""
    import test_host_pnp as hp
    from oracle import ref_py
    from pvio_amd.problem import BAProblem
    _d = ref_py._d
    pb, T, Lf, fac = hp.make_case(use_inertial)
    x0 = pb.frame_state[T].copy()
    x0[4:7] += [0.05, -0.04, 0.03]
    d, tmp = np.zeros(15), np.zeros(16)
    d[0:3] = [0.02, -0.015, 0.01]
    oracle.lib().oracle_plus(_d(np.ascontiguousarray(x0)), _d(d), _d(tmp))
    x0[0:4] = tmp[0:4]
    win = BAProblem(T)
    for name in ("frame_fixed", "cam_extrinsic", "imu_extrinsic", "sqrt_inv_cov", "intrinsics"):
        setattr(win, name, getattr(pb, name)[:T].copy())
    win.max_iterations = 10
    ptr, frame, z, rho, index = [0], [], [], [], {}
    for l in range(pb.n_landmarks):
        a = int(pb.lm_anchor_frame[l])
        if a >= T:
            continue
        frame.append(a), z.append(pb.lm_anchor_z[l])
        for o in range(pb.lm_obs_ptr[l], pb.lm_obs_ptr[l + 1]):
            if pb.obs_frame[o] < T:
                frame.append(int(pb.obs_frame[o])), z.append(pb.obs_z[o])
        index[l] = len(ptr) - 1
        ptr.append(len(frame)), rho.append(pb.lm_inv_depth[l])
    nt = len(ptr) - 1
    mk = lambda: ref.Tracks(ptr, frame, np.array(z), np.array(rho), np.ones(nt, np.uint8), np.zeros(nt, np.uint8))  # noqa: E731
    obs_track = [index[l] for (l, _, _) in fac]
    obs_z = np.array([pb.obs_z[o] for (_, _, o) in fac])
    args = (pb.cam_extrinsic[T], pb.imu_extrinsic[T], pb.sqrt_inv_cov[T], pb.intrinsics[T], obs_track, obs_z, pb.preint_delta[T], pb.preint_sqrt_inv_cov[T],
            pb.preint_jacobian[T], use_inertial)
    xa, _ = ref.reference().pnp(win, pb.frame_state[:T], mk(), x0, *args)
    xb, _ = drop.pnp(win, pb.frame_state[:T], mk(), x0, *args)
    na = 16 if use_inertial else 7
    assert np.abs(xb[:na] - xa[:na]).max() < 1e-8, np.abs(xb[:na] - xa[:na]).max()
    assert np.abs(xa[:7] - x0[:7]).max() > 1e-3  # and the call moved the pose
    return dict(use_inertial=use_inertial, worst=float(np.abs(xb[:na] - xa[:na]).max()))


@pytest.mark.parametrize("use_inertial", [False, True])
def test_dropin_pnp_on_reference_map(ref, emu, oracle, use_inertial):  # noqa: F811
    """host only (pvio_amd/host/pnp.cpp + pnp_solve.cpp make no device call), so the emulated library is the product here"""
    print(diff_pnp(ref, emu, oracle, use_inertial))


# ---- the same on the MI355X: libpvio_dropin.so above libpvio_hip.so ------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SOLVE_CASES) + list(BIG))
def test_gpu_dropin_solve_on_reference_map(ref, gpu, oracle, name):  # noqa: F811
    """incl. the window the metric is quoted on: the reference's solve of it goes through mini-Ceres' dense Cholesky of 1150 unknowns (seconds)"""
    pb = ba_compare.make(oracle, **dict(SOLVE_CASES, **BIG)[name])
    print(name, diff_solve(ref, gpu, pb))


@pytest.mark.gpu
def test_gpu_dropin_post_solve_passes_on_reference_map(ref, gpu, oracle):  # noqa: F811
    print(_post_pass_diff(ref, gpu, oracle))


@pytest.mark.gpu
@pytest.mark.parametrize("victim", [0, 3])
def test_gpu_dropin_marginalize_on_reference_map(ref, gpu, oracle, victim):  # noqa: F811
    print(diff_marginalize(ref, gpu, oracle, victim, regular_prior=(victim != 0)))


@pytest.mark.gpu
def test_gpu_dropin_keyframe_cycle_on_reference_map(ref, gpu, oracle):  # noqa: F811
    print(diff_cycle(ref, gpu, oracle, n_frames=7, n_landmarks=120, use_inertial=True, visibility=5))
    print(diff_cycle(ref, gpu, oracle, n_frames=10, n_landmarks=300, use_inertial=True, visibility=6))


def _random_window(oracle, seed):
    """the shapes of tests/sweep_random_dropin.py"""
    rng = np.random.default_rng(8000 + seed)
    n = int(rng.integers(3, 13))
    kw = dict(n_frames=n, n_landmarks=int(rng.integers(20, 400)), use_inertial=bool(rng.integers(0, 2)), visibility=int(rng.integers(2, n + 1)),
              plane_fraction=float(rng.choice([0.0, 0.0, 0.3, 0.5])), seed=int(rng.integers(1, 10000)))
    if rng.random() < 0.25:
        kw["duplicate_fraction"] = 0.3
        kw["plane_fraction"] = 0.0
    return ba_compare.make(oracle, **kw)


@pytest.mark.parametrize("seed", [5, 9, 18])
def test_emulated_dropin_random_windows_on_reference_map(ref, emu, oracle, seed):  # noqa: F811
    print(seed, diff_solve(ref, emu, _random_window(oracle, seed)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 3, 5, 7, 9, 11, 14, 16, 18, 22])
def test_gpu_dropin_random_windows_on_reference_map(ref, gpu, oracle, seed):  # noqa: F811
    print(seed, diff_solve(ref, gpu, _random_window(oracle, seed)))
