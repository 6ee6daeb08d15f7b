"""Checks of a C-ABI context (emulated build or the real GPU library) or of the oracle against tests/golden/*.npz."""
import numpy as np

import golden_io
from pvio_amd import BAState, BASummary

BA_FIXTURES = ["ba_vision_4x30", "ba_vio_4x30", "ba_vio_plane_5x60", "ba_vio_partial_6x40"]
MARG_FIXTURES = ["marg_vio_4x30_victim0", "marg_vio_partial_6x40_victim0"]
FRONT_FIXTURE = "front_176x132"


def ba_oracle(oracle, name):
    d = golden_io.load(name + ".npz")
    pb = golden_io.problem_from_dict(d)
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)
    # same code, same inputs: anything beyond last-bit libm differences is a behaviour change of the oracle
    return golden_io.check_solution(d, st, sm, state_tol=1e-12, cost_rtol=1e-12)


def ba_ctx(ctx, name, state_tol=1e-6):
    d = golden_io.load(name + ".npz")
    pb = golden_io.problem_from_dict(d)
    st, sm = ctx.solve(pb)
    return golden_io.check_solution(d, st, sm, state_tol=state_tol, cost_rtol=1e-7)


def _marg_inputs(name):
    d = golden_io.load(name + ".npz")
    pb = golden_io.problem_from_dict(d)
    st = BAState(pb)
    st.frame_state[:] = d["out_frame_state"]
    st.lm_inv_depth[:] = d["out_lm_inv_depth"]
    st.lm_valid[:] = d["out_lm_valid"]
    return d, pb, st, int(d["victim"][0])


def marg(solver, name, rtol):
    """solver: the oracle module or a HipContext (both expose marginalize(problem, state, victim))."""
    d, pb, st, victim = _marg_inputs(name)
    S, s, IM, iv = solver.marginalize(pb, st, victim)
    scale = np.abs(d["out_info_matrix"]).max()
    np.testing.assert_allclose(IM, d["out_info_matrix"], rtol=rtol, atol=rtol * 1e-2 * scale)
    np.testing.assert_allclose(iv, d["out_info_vector"], rtol=rtol, atol=rtol * 1e-2 * np.abs(d["out_info_vector"]).max())
    np.testing.assert_allclose(S.T @ S, d["out_StS"], rtol=10 * rtol, atol=10 * rtol * scale)
    np.testing.assert_allclose(S.T @ s, d["out_Sts"], rtol=10 * rtol, atol=10 * rtol * np.abs(d["out_Sts"]).max())


def front_oracle(oracle):
    d = golden_io.load(FRONT_FIXTURE + ".npz")
    c0, c1 = oracle.clahe(d["in_img0"]), oracle.clahe(d["in_img1"])
    assert (c0 == d["out_clahe0"]).all()
    P0, P1 = oracle.build_pyramid(c0), oracle.build_pyramid(c1)
    for l in range(len(P0)):
        if l > 0:
            assert (P0[l][0] == d["out_level%d_image" % l]).all()
        assert (P0[l][1] == d["out_level%d_deriv" % l]).all()
    nxt, status = oracle.klt_track(P0, P1, d["in_prev_xy"], d["in_init_xy"])
    assert (status == d["out_status"]).all()
    assert (nxt == d["out_next_xy"])[status > 0].all()   # the defined order of the float sums (oracle_klt.cpp header); no libm call rounds differently
    nxt_s, status_s = oracle.klt_track(P0, P1, d["in_prev_xy"], d["in_init_xy"], scalar_order=True)
    assert (status_s == d["out_status"]).all() and (nxt_s == d["out_next_xy_scalar_order"])[status > 0].all()
    assert np.abs(nxt - nxt_s)[status > 0].max() <= 1e-3        # OpenCV's scalar order: the same tracker up to the rounding of 441-term float sums
    resp = oracle.harris_response(c0)
    assert (resp.view(np.int32) == d["out_harris"].view(np.int32)).all()
    mc, q, md = d["in_detect_params"]
    xy, r = oracle.good_features(resp, int(mc), float(q), float(md))
    assert (xy == d["out_corners_xy"]).all() and (r.view(np.int32) == d["out_corners_resp"].view(np.int32)).all()


def front_ctx(ctx, pos_tol=0.0):
    from pvio_amd.solver import HipImage, detect_corners, klt_track
    d = golden_io.load(FRONT_FIXTURE + ".npz")
    A, B = HipImage(ctx, d["in_img0"], True), HipImage(ctx, d["in_img1"], True)
    n_levels = 1 + sum(1 for k in d if k.startswith("out_level") and k.endswith("_image"))
    for l in range(n_levels):
        gi, gd = A.level(l)
        want = d["out_clahe0"] if l == 0 else d["out_level%d_image" % l]
        assert (gi == want).all(), "level %d image differs" % l
        assert (gd == d["out_level%d_deriv" % l]).all(), "level %d derivative differs" % l
    nxt, status, _ = klt_track(ctx, A, B, d["in_prev_xy"], d["in_init_xy"])
    assert (status == d["out_status"]).all()
    assert np.abs(nxt - d["out_next_xy"])[status > 0].max() <= pos_tol  # 0: bit-identical positions (the kernel sums in the oracle's defined order)
    mc, q, md = d["in_detect_params"]
    xy, r, rmap = detect_corners(ctx, A, int(mc), float(q), float(md), want_response_map=True)
    assert (rmap.view(np.int32) == d["out_harris"].view(np.int32)).all()
    assert (xy == d["out_corners_xy"]).all() and (r.view(np.int32) == d["out_corners_resp"].view(np.int32)).all()
    A.release()
    B.release()


def undistort_oracle(oracle):
    """The numpy restatement (and the oracle's CLAHE behind it) against the committed maps and pixels."""
    from oracle import oracle_undistort as U
    g = np.load(golden_io.path("undistort_small.npz"))
    he, we = g["in_euroc_src"].shape
    xy, fr = U.cv_undistort_fixed_maps(g["in_euroc_K"], g["in_euroc_dist"], we, he)
    assert (xy == g["out_euroc_map_xy"]).all() and (fr == g["out_euroc_map_frac"]).all()
    ht, wt = g["in_tum_src"].shape
    xy, fr = U.image_undistorter_maps(wt, ht, g["in_tum_K"], g["in_tum_dist"], "equidistant")
    assert (xy == g["out_tum_map_xy"]).all() and (fr == g["out_tum_map_frac"]).all()
    for cam in ("euroc", "tum"):
        out = U.remap_bilinear(g["in_%s_src" % cam], g["out_%s_map_xy" % cam], g["out_%s_map_frac" % cam])
        assert (out == g["out_%s_remap" % cam]).all()
        assert (oracle.clahe(out) == g["out_%s_level0" % cam]).all()


def undistort_ctx(ctx):
    """The device remap (+ CLAHE) on the committed maps against the committed pixels."""
    from pvio_amd.solver import HipImage, HipUndistort
    g = np.load(golden_io.path("undistort_small.npz"))
    for cam in ("euroc", "tum"):
        ud = HipUndistort(ctx, g["out_%s_map_xy" % cam], g["out_%s_map_frac" % cam])
        plain = HipImage(ctx, g["in_%s_src" % cam], clahe=False, undistort=ud)
        assert (plain.level(0)[0] == g["out_%s_remap" % cam]).all()
        eq = HipImage(ctx, g["in_%s_src" % cam], clahe=True, undistort=ud)
        assert (eq.level(0)[0] == g["out_%s_level0" % cam]).all()
        plain.release(), eq.release(), ud.release()
