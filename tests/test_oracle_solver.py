"""The C++ oracle's trust-region loop vs the independent dense numpy implementation (tests/np_reference.py),
plus solver-level invariants.  PARITY UNPINNED against real Ceres (not available) -- see oracle/README.md."""
import numpy as np
import pytest

import np_reference
from pvio_amd import BAState, BASummary, synth

CASES = {
    "vision": dict(n_frames=4, n_landmarks=30),
    "vision_partial": dict(n_frames=6, n_landmarks=40, visibility=3),
    "vio": dict(n_frames=4, n_landmarks=30, use_inertial=True),
    "vio_partial": dict(n_frames=6, n_landmarks=40, use_inertial=True, visibility=4),
    "plane": dict(n_frames=5, n_landmarks=60, plane_fraction=0.5),
    "vio_plane": dict(n_frames=5, n_landmarks=60, plane_fraction=0.4, use_inertial=True),
    "vision_rot_prior": dict(n_frames=5, n_landmarks=40, visibility=4, rot_prior_frames=(0, 2, 4)),
    "vio_rot_prior": dict(n_frames=5, n_landmarks=40, use_inertial=True, visibility=4, rot_prior_frames=(1, 3, 4)),
}


def make(oracle, **kw):
    if kw.get("use_inertial"):
        kw["preintegrate"] = oracle.preintegrate
    return synth.make_window(**kw)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_dense_numpy(oracle, name):
    pb = make(oracle, **CASES[name])
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)
    trace, fs, rho, term, iters = np_reference.solve(pb, oracle)
    otrace = sm.trace()
    assert sm.termination == term
    assert sm.num_iterations == iters
    assert len(otrace) == len(trace)
    for a, b in zip(otrace, trace):
        assert a["iteration"] == b["iteration"]
        assert a["step_is_successful"] == b["step_is_successful"], (a, b)
        assert a["step_is_valid"] == b["step_is_valid"]
        np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-7)
        np.testing.assert_allclose(a["trust_region_radius"], b["trust_region_radius"], rtol=1e-7)
        np.testing.assert_allclose(a["mu"], b["mu"], rtol=1e-12)
        np.testing.assert_allclose(a["step_norm"], b["step_norm"], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(a["gradient_max_norm"], b["gradient_max_norm"], rtol=1e-6, atol=1e-9)
    for k, b in enumerate(trace):
        np.testing.assert_allclose(sm.trace_states[k], b["state"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(st.frame_state, fs, atol=1e-7)
    np.testing.assert_allclose(st.lm_inv_depth, rho, atol=1e-7)


def test_schur_equals_dense_normal_equations(oracle):
    """Hpp/W/Hll from the oracle reproduce J^T J of the dense Jacobian."""
    pb = make(oracle, **CASES["vio_partial"])
    lin = oracle.linearize(pb, pb.frame_state, pb.lm_inv_depth)
    D = np_reference.DenseProblem(pb, oracle)
    cost, r, J = D.evaluate(pb.frame_state, pb.lm_inv_depth, pb.frame_state)
    H = J.T @ J
    g = J.T @ r
    P = lin["P"]
    scale = np.abs(H[:P, :P]).max()
    np.testing.assert_allclose(lin["Hpp"], H[:P, :P], rtol=1e-9, atol=1e-12 * scale)
    np.testing.assert_allclose(lin["gp"], g[:P], rtol=1e-9, atol=1e-9 * np.abs(g).max())
    np.testing.assert_allclose(lin["Hll"], np.diag(H)[P:], rtol=1e-10)
    np.testing.assert_allclose(lin["bl"], g[P:], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(lin["W"], H[P:, :P], rtol=1e-9, atol=1e-9 * np.abs(H[P:, :P]).max())
    np.testing.assert_allclose(lin["cost"], cost, rtol=1e-12)


def test_zero_noise_truth_is_a_fixed_point(oracle):
    """Noise-free observations + truth initial guess: zero reprojection residual, solver stops immediately."""
    pb = synth.make_window(n_frames=4, n_landmarks=25, perturb=False)
    # regenerate noise-free observations from the truth geometry
    pts = pb.meta["points"]
    R_bc = synth.qmat(pb.cam_extrinsic[0, :4])
    for l in range(pb.n_landmarks):
        for o in range(pb.lm_obs_ptr[l], pb.lm_obs_ptr[l + 1]):
            f = pb.obs_frame[o]
            Rw = synth.qmat(pb.truth_frame_state[f, :4]) @ R_bc
            pw = pb.truth_frame_state[f, 4:7] + synth.qmat(pb.truth_frame_state[f, :4]) @ pb.cam_extrinsic[f, 4:7]
            y = Rw.T @ (pts[l] - pw)
            pb.obs_z[o] = y[:2] / y[2]
        a = pb.lm_anchor_frame[l]
        Rw = synth.qmat(pb.truth_frame_state[a, :4]) @ R_bc
        pw = pb.truth_frame_state[a, 4:7] + synth.qmat(pb.truth_frame_state[a, :4]) @ pb.cam_extrinsic[a, 4:7]
        y = Rw.T @ (pts[l] - pw)
        pb.lm_anchor_z[l] = y[:2] / y[2]
        pb.lm_inv_depth[l] = 1.0 / y[2]
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)
    assert sm.initial_cost < 1e-18
    np.testing.assert_allclose(st.frame_state, pb.frame_state, atol=1e-9)
    assert st.lm_quality.max() < 1e-6


def test_metric_config_converges_and_reduces_cost(oracle):
    pb = synth.make_window(n_frames=10, n_landmarks=200)
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)
    assert sm.is_usable == 1
    assert sm.final_cost < 0.3 * sm.initial_cost
    assert (st.lm_valid == 1).all()
    assert 0.5 < st.lm_quality.mean() < 1.5  # ~ sqrt(pi/2 * 0.5) px for 0.5 px^2 noise
    tr = sm.trace()
    costs = [t["cost"] for t in tr if t["step_is_successful"]]
    assert all(b < a for a, b in zip(costs, costs[1:]))


@pytest.mark.parametrize("case,victim", [("vio", 0), ("vio", 1), ("vio", 3), ("vio_partial", 0), ("vio_partial", 2), ("vio_rot_prior", 1), ("vio_rot_prior", 2),
                                         ("vio_partial", 5), ("vio_plane", 0), ("vio_plane", 4)])
def test_oracle_marginalization_matches_dense_numpy(oracle, case, victim):
    """marginalize_frame (bundle_adjustor.cpp:348-599): the oracle's block accumulation + two-stage elimination against
    the Schur complement of a dense J^T J (np_reference.marginalize)."""
    import marg_compare
    pb, st = marg_compare.solved_window(oracle, regular_prior=(victim != 0), **CASES[case])
    S, s, IM, iv = oracle.marginalize(pb, st, victim)
    IM2, iv2 = np_reference.marginalize(pb, oracle, st.frame_state, st.lm_inv_depth, victim)
    scale = np.abs(IM).max()
    np.testing.assert_allclose(IM, IM2, rtol=0, atol=1e-11 * scale)
    np.testing.assert_allclose(iv, iv2, rtol=0, atol=1e-11 * np.abs(iv).max())
    # the square-root form reproduces both (eigenvalues <= 1e-8 dropped, :583-590)
    np.testing.assert_allclose(S.T @ S, IM2, rtol=0, atol=1e-7 * scale)
