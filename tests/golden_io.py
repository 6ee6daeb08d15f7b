"""Self-contained fixtures under tests/golden/: the complete inputs of a window (or an image pair) together with the
outputs the CPU oracle produced for them, written by tests/golden/make_golden.py.

The reference repository ships no golden vectors for this path and neither Ceres nor OpenCV can be
built here, so these files do NOT pin the oracle to the reference ("parity unpinned" stays true).  What they do:
freeze the oracle's answers at the moment two independent restatements (C++ oracle, dense numpy loop) agreed on them,
so that a later change to either the oracle or the HIP path that moves a result shows up against a committed file.
"""
import json
import os

import numpy as np

from pvio_amd.problem import BAProblem

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PROBLEM_ARRAYS = ["frame_fixed", "cam_extrinsic", "imu_extrinsic", "sqrt_inv_cov", "intrinsics", "lm_anchor_frame",
                  "lm_anchor_z", "lm_obs_ptr", "obs_frame", "obs_z", "preint_valid", "preint_delta",
                  "preint_sqrt_inv_cov", "preint_jacobian", "prior_frames", "prior_S", "prior_s", "prior_lin_state",
                  "plane_obs_ptr", "plane_obs_frame", "plane_obs_z", "plane_normal", "plane_distance", "frame_state",
                  "lm_inv_depth"]
PROBLEM_SCALARS = ["use_inertial", "plane_sqrt_inv_cov", "max_iterations", "max_solver_time"]
TRACE_FIELDS = ["iteration", "step_is_valid", "step_is_successful", "cost", "cost_change", "gradient_max_norm",
                "step_norm", "relative_decrease", "trust_region_radius", "mu"]


def path(name):
    return os.path.join(GOLDEN_DIR, name)


def problem_to_dict(pb):
    pb._canon()
    d = {"in_" + k: getattr(pb, k) for k in PROBLEM_ARRAYS}
    d["in_scalars"] = np.frombuffer(json.dumps({k: getattr(pb, k) for k in PROBLEM_SCALARS}).encode(), np.uint8)
    return d


def problem_from_dict(d):
    pb = BAProblem(int(d["in_frame_fixed"].shape[0]))
    for k in PROBLEM_ARRAYS:
        setattr(pb, k, np.array(d["in_" + k]))
    for k, v in json.loads(bytes(d["in_scalars"]).decode()).items():
        setattr(pb, k, v)
    pb._canon()
    return pb


def solution_to_dict(state, summary):
    tr = summary.trace()
    d = {"out_" + f: np.array([t[f] for t in tr]) for f in TRACE_FIELDS if f in tr[0]}
    d["out_trace_states"] = summary.trace_states[:len(tr)].copy()
    d["out_frame_state"] = state.frame_state.copy()
    d["out_lm_inv_depth"] = state.lm_inv_depth.copy()
    d["out_lm_valid"] = state.lm_valid.copy()
    d["out_lm_quality"] = state.lm_quality.copy()
    d["out_summary"] = np.array([summary.termination, summary.is_usable, summary.num_iterations,
                                 summary.num_successful_steps], np.int64)
    d["out_costs"] = np.array([summary.initial_cost, summary.final_cost])
    return d


def check_solution(d, state, summary, state_tol, cost_rtol=1e-9):
    """Compare a solve (oracle or C-ABI) with the frozen outputs."""
    tr = summary.trace()
    assert [summary.termination, summary.is_usable, summary.num_iterations, summary.num_successful_steps] == \
        list(d["out_summary"])
    assert len(tr) == len(d["out_iteration"])
    for f in ("iteration", "step_is_valid", "step_is_successful"):
        assert [t[f] for t in tr] == list(d["out_" + f]), f
    np.testing.assert_allclose([t["cost"] for t in tr], d["out_cost"], rtol=cost_rtol)
    np.testing.assert_allclose([t["mu"] for t in tr], d["out_mu"], rtol=1e-12)
    np.testing.assert_allclose([t["trust_region_radius"] for t in tr], d["out_trust_region_radius"], rtol=1e-6)
    np.testing.assert_allclose(summary.trace_states[:len(tr)], d["out_trace_states"], rtol=0, atol=state_tol)
    np.testing.assert_allclose(state.frame_state, d["out_frame_state"], rtol=0, atol=state_tol)
    np.testing.assert_allclose(state.lm_inv_depth, d["out_lm_inv_depth"], rtol=0, atol=state_tol)
    assert (state.lm_valid == d["out_lm_valid"]).all()
    np.testing.assert_allclose(state.lm_quality, d["out_lm_quality"], rtol=0, atol=1e-5)
    np.testing.assert_allclose([summary.initial_cost, summary.final_cost], d["out_costs"], rtol=cost_rtol)
    return float(np.abs(summary.trace_states[:len(tr)] - d["out_trace_states"]).max())


def load(name):
    with np.load(path(name)) as z:
        return {k: z[k] for k in z.files}
