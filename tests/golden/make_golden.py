#!/usr/bin/env python3
"""Writes the fixtures in this directory.  Run from the repository root:  python tests/golden/make_golden.py

Every file holds the full inputs and the CPU oracle's outputs for them (see tests/golden_io.py for what that does and
does not prove).  A fixture is only written after the second, independent restatement agreed with the oracle:
 - bundle adjustment: the dense numpy trust-region loop of tests/np_reference.py (trace, states after every iteration);
 - marginalization: np_reference.marginalize (Schur complement of a dense J^T J);
 - image front end: the known-answer properties of tests/test_oracle_klt.py hold for the same oracle build (checked
   by the CPU suite); the pair itself has no second implementation, the file only freezes the oracle.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))           # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root

import golden_io  # noqa: E402
import np_reference  # noqa: E402
from oracle import oracle_py as oracle  # noqa: E402
from pvio_amd import BAState, BASummary, synth  # noqa: E402

BA_CASES = {
    "ba_vision_4x30": dict(n_frames=4, n_landmarks=30),
    "ba_vio_4x30": dict(n_frames=4, n_landmarks=30, use_inertial=True),
    "ba_vio_plane_5x60": dict(n_frames=5, n_landmarks=60, plane_fraction=0.4, use_inertial=True),
    "ba_vio_partial_6x40": dict(n_frames=6, n_landmarks=40, use_inertial=True, visibility=4),
}
MARG_CASES = {
    "marg_vio_4x30_victim0": (dict(n_frames=4, n_landmarks=30, use_inertial=True), 0),
    "marg_vio_partial_6x40_victim0": (dict(n_frames=6, n_landmarks=40, use_inertial=True, visibility=4), 0),
}


def make_window(**kw):
    if kw.get("use_inertial"):
        kw = dict(kw, preintegrate=oracle.preintegrate)
    return synth.make_window(**kw)


def write_ba(name, kw):
    pb = make_window(**kw)
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)
    trace, fs, rho, term, iters = np_reference.solve(pb, oracle)
    assert sm.termination == term and sm.num_iterations == iters and len(trace) == sm.trace_len
    for k, b in enumerate(trace):
        np.testing.assert_allclose(sm.trace_states[k], b["state"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(sm.trace()[k]["cost"], b["cost"], rtol=1e-7)
    d = golden_io.problem_to_dict(pb)
    d.update(golden_io.solution_to_dict(st, sm))
    np.savez_compressed(golden_io.path(name + ".npz"), **d)
    print(name, "iterations", sm.num_iterations, "final cost %.9g" % sm.final_cost)


def write_marg(name, kw, victim):
    pb = make_window(**kw)
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)
    S, s, IM, iv = oracle.marginalize(pb, st, victim)
    IM2, iv2 = np_reference.marginalize(pb, oracle, st.frame_state, st.lm_inv_depth, victim)
    np.testing.assert_allclose(IM, IM2, rtol=0, atol=1e-11 * np.abs(IM).max())
    np.testing.assert_allclose(iv, iv2, rtol=0, atol=1e-11 * np.abs(iv).max())
    d = golden_io.problem_to_dict(pb)
    d.update(out_frame_state=st.frame_state, out_lm_inv_depth=st.lm_inv_depth, out_lm_valid=st.lm_valid,
             victim=np.array([victim]), out_info_matrix=IM, out_info_vector=iv, out_StS=S.T @ S, out_Sts=S.T @ s)
    np.savez_compressed(golden_io.path(name + ".npz"), **d)
    print(name, "prior dimension", S.shape)


def write_front(name, width, height, n_points):
    img0, img1, p, truth, init = synth.make_image_pair(width, height, n_points)
    c0, c1 = oracle.clahe(img0), oracle.clahe(img1)
    P0, P1 = oracle.build_pyramid(c0), oracle.build_pyramid(c1)
    nxt, status = oracle.klt_track(P0, P1, p, init)  # the float sums in the defined order (oracle_klt.cpp header): the bit-exact contract
    nxt_scalar, status_scalar = oracle.klt_track(P0, P1, p, init, scalar_order=True)  # OpenCV's scalar left-to-right order
    assert (status == status_scalar).all()
    resp = oracle.harris_response(c0)
    xy, r = oracle.good_features(resp, 200, 1.0e-3, 12.0)
    d = dict(in_img0=img0, in_img1=img1, in_prev_xy=p, in_init_xy=init, in_truth_xy=truth,
             out_clahe0=c0, out_next_xy=nxt, out_next_xy_scalar_order=nxt_scalar, out_status=status, out_harris=resp, out_corners_xy=xy, out_corners_resp=r,
             in_detect_params=np.array([200, 1.0e-3, 12.0]))
    for l in range(1, len(P0)):          # level 0 image is the CLAHE output; its derivative and the rest are stored as is
        d["out_level%d_image" % l] = P0[l][0]
    for l in range(len(P0)):
        d["out_level%d_deriv" % l] = P0[l][1]
    np.savez_compressed(golden_io.path(name + ".npz"), **d)
    print(name, "tracks kept", int((status > 0).sum()), "of", len(status), "corners", len(xy))


def write_undistort(name):
    """The dataset readers' undistortion (oracle/oracle_undistort.py): maps of both camera models at a small size with the
    reference readers' constants scaled down, a random source image, the remap output and its CLAHE'd level 0."""
    from oracle import oracle_undistort as U
    rng = np.random.default_rng(77)
    s = 0.25                                                     # EuRoC camera at a quarter of its resolution
    K_e = [458.654 * s, 0, 367.215 * s, 0, 457.296 * s, 248.375 * s, 0, 0, 1]
    D_e = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]
    we, he = 188, 120
    xy_e, fr_e = U.cv_undistort_fixed_maps(K_e, D_e, we, he)
    K_t = [190.97847715128717 * s, 0, 254.93170605935475 * s, 0, 190.9733070521226 * s, 256.8974428996504 * s, 0, 0, 1]
    D_t = [0.0034003170790442797, 0.001766278153469831, -0.00266312569781606, 0.0003299517423931039]
    wt, ht = 128, 128
    xy_t, fr_t = U.image_undistorter_maps(wt, ht, K_t, D_t, "equidistant")
    src_e = rng.integers(0, 256, (he, we), dtype=np.uint8)
    src_t = rng.integers(0, 256, (ht, wt), dtype=np.uint8)
    out_e, out_t = U.remap_bilinear(src_e, xy_e, fr_e), U.remap_bilinear(src_t, xy_t, fr_t)
    d = dict(in_euroc_K=np.array(K_e), in_euroc_dist=np.array(D_e), in_euroc_src=src_e, out_euroc_map_xy=xy_e, out_euroc_map_frac=fr_e,
             out_euroc_remap=out_e, out_euroc_level0=oracle.clahe(out_e),
             in_tum_K=np.array(K_t), in_tum_dist=np.array(D_t), in_tum_src=src_t, out_tum_map_xy=xy_t, out_tum_map_frac=fr_t,
             out_tum_remap=out_t, out_tum_level0=oracle.clahe(out_t))
    np.savez_compressed(golden_io.path(name + ".npz"), **d)
    print(name, "euroc corner ->", xy_e[0, 0], "tum corner ->", xy_t[0, 0])


if __name__ == "__main__":
    for n, kw in BA_CASES.items():
        write_ba(n, kw)
    for n, (kw, v) in MARG_CASES.items():
        write_marg(n, kw, v)
    write_front("front_176x132", 176, 132, 60)
    write_undistort("undistort_small")
