"""Third-party pins for the CPU oracle (scipy is in the image; Ceres / OpenCV / Eigen are not).

The oracle restates the reference's arithmetic and cannot be checked against the reference itself (profiles/NOTES_r1_r3.md section 2:
parity unpinned).  Where an INDEPENDENT third-party implementation of the same mathematical object exists in this image it
is used here as a pin -- not a pin to Ceres or OpenCV, but to code nobody in this repository wrote:

  * SO(3) exponential / logarithm                      scipy.spatial.transform.Rotation
  * pyrDown's separable [1 4 6 4 1] kernel, reflect-101 scipy.ndimage.correlate1d(mode="mirror")
  * Scharr derivative planes (interior)                scipy.ndimage.correlate
  * Harris response (interior, float tolerance)        scipy.ndimage Sobel / uniform filters in float64
  * the robustified bundle-adjustment OBJECTIVE        scipy.optimize.least_squares started at the oracle's solution cannot
                                                       lower it, and evaluates it to the value the oracle reports
"""
import ctypes as C

import numpy as np
import pytest
from scipy import ndimage, optimize
from scipy.spatial.transform import Rotation

import np_reference
from pvio_amd import BAState, BASummary, synth

dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(dp)


def test_lie_helpers_against_scipy_rotation(oracle):
    rng = np.random.default_rng(21)
    L = oracle.lib()
    for _ in range(200):
        w = rng.normal(size=3) * rng.choice([1e-7, 1e-3, 0.5, 2.0])
        if np.linalg.norm(w) > 3.0:
            w *= 3.0 / np.linalg.norm(w)
        q = np.zeros(4)
        L.oracle_expmap(_d(w), _d(q))
        qs = Rotation.from_rotvec(w).as_quat()  # x y z w, like Eigen's coeffs()
        if qs[3] < 0:
            qs = -qs
        np.testing.assert_allclose(q if q[3] >= 0 else -q, qs, rtol=0, atol=1e-15)
        w2 = np.zeros(3)
        L.oracle_logmap(_d(np.ascontiguousarray(qs)), _d(w2))
        np.testing.assert_allclose(w2, Rotation.from_quat(qs).as_rotvec(), rtol=1e-12, atol=1e-15)
        # right Jacobian: Exp(w + d) = Exp(w) Exp(Jr d) + O(d^2), the product evaluated by scipy
        Jr = np.zeros(9)
        L.oracle_right_jacobian(_d(w), _d(Jr))
        d = rng.normal(size=3) * 1e-6
        lhs = (Rotation.from_rotvec(w).inv() * Rotation.from_rotvec(w + d)).as_rotvec()
        np.testing.assert_allclose(lhs, Jr.reshape(3, 3) @ d, rtol=2e-4, atol=1e-11)


def test_pyr_down_is_scipy_separable_filter(oracle):
    rng = np.random.default_rng(22)
    for (h, w) in ((64, 96), (75, 101), (48, 48)):
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
        lv = oracle.build_pyramid(img)
        k = np.array([1, 4, 6, 4, 1], np.int64)
        f = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.int64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")  # mirror = BORDER_REFLECT_101
        want = ((f + 128) >> 8)[::2, ::2].astype(np.uint8)
        assert lv[1][0].shape == ((h + 1) // 2, (w + 1) // 2)
        assert (lv[1][0] == want).all()


def test_scharr_planes_are_scipy_correlations_in_the_interior(oracle):
    rng = np.random.default_rng(23)
    img = rng.integers(0, 256, (60, 80)).astype(np.uint8)
    d = oracle.build_pyramid(img)[0][1].astype(np.int64)
    smooth, diff = np.array([3, 10, 3], np.int64), np.array([-1, 0, 1], np.int64)
    I = img.astype(np.int64)
    dx = ndimage.correlate1d(ndimage.correlate1d(I, diff, axis=1, mode="mirror"), smooth, axis=0, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(I, diff, axis=0, mode="mirror"), smooth, axis=1, mode="mirror")
    assert (d[1:-1, 1:-1, 0] == dx[1:-1, 1:-1]).all() and (d[1:-1, 1:-1, 1] == dy[1:-1, 1:-1]).all()


def test_harris_response_against_scipy_filters(oracle):
    """cornerHarris(blockSize 3, ksize 3, k 0.04): Sobel gradients scaled by 1 / (4 * 3 * 255), unnormalized 3 x 3 box sums of the
    products, det - k trace^2 -- recomputed in float64 with scipy's filters; the oracle's float32 map must agree to float32
    accuracy (the reflect-101 border is the same in both, so the whole map is compared)."""
    rng = np.random.default_rng(24)
    img = synth.make_image_pair(160, 120, 8)[0]
    img = np.clip(img.astype(int) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    r = oracle.harris_response(img).astype(np.float64)
    I = img.astype(np.float64)
    s = 1.0 / (4.0 * 3.0 * 255.0)
    dx = s * ndimage.correlate1d(ndimage.correlate1d(I, [-1, 0, 1], axis=1, mode="mirror"), [1, 2, 1], axis=0, mode="mirror")
    dy = s * ndimage.correlate1d(ndimage.correlate1d(I, [-1, 0, 1], axis=0, mode="mirror"), [1, 2, 1], axis=1, mode="mirror")
    box = lambda a: ndimage.correlate(a, np.ones((3, 3)), mode="mirror")
    a, b, c = box(dx * dx), box(dx * dy), box(dy * dy)
    want = a * c - b * b - 0.04 * (a + c) ** 2
    np.testing.assert_allclose(r, want, rtol=0, atol=2e-6 * np.abs(want).max())
    assert np.corrcoef(r.ravel(), want.ravel())[0, 1] > 0.999999


class _Objective:
    """0.5 * |f(delta)|^2 = the reference's objective 0.5 * sum_b rho_b(|r_b|^2) at x0 (+) delta: rows r_b * sqrt(rho(s) / s) for the
    Cauchy-robustified blocks (s = |r_b|^2), r_b itself for the others -- built on tests/np_reference.DenseProblem, whose rows are
    the Ceres-corrected ones (r_b * sqrt(rho'(s))): the block structure is recovered from its bookkeeping."""

    def __init__(self, pb, oracle):
        self.D = np_reference.DenseProblem(pb, oracle)
        self.pb = pb
        self.fs0, self.rho0 = pb.frame_state.copy(), pb.lm_inv_depth.copy()

    def state(self, delta):
        return self.D.plus(self.fs0, self.rho0, delta)

    def rows(self, delta):
        fs, rho = self.state(delta)
        pb = self.pb
        out = []
        L = self.D.L
        for l in range(pb.n_landmarks):
            a = pb.lm_anchor_frame[l]
            for o in range(pb.lm_obs_ptr[l], pb.lm_obs_ptr[l + 1]):
                t = pb.obs_frame[o]
                r = np.zeros(2)
                L.oracle_eval_reprojection(_d(np.ascontiguousarray(fs[t])), _d(np.ascontiguousarray(fs[a])), float(rho[l]), _d(pb.lm_anchor_z[l]), _d(pb.obs_z[o]),
                                           _d(pb.cam_extrinsic[a]), _d(pb.cam_extrinsic[t]), _d(pb.sqrt_inv_cov[t]), _d(r), None)
                s = float(r @ r)
                out.append(r * (np.sqrt(np.log1p(s) / s) if s > 0 else 1.0))
        for f in range(pb.n_plane_factors):
            b, e = pb.plane_obs_ptr[f], pb.plane_obs_ptr[f + 1]
            frames = pb.plane_obs_frame[b:e]
            if all(pb.frame_fixed[frames]):
                continue
            r = np.zeros(1)
            L.oracle_eval_plane(int(e - b), _d(np.ascontiguousarray(fs[frames])), _d(np.ascontiguousarray(pb.cam_extrinsic[frames])),
                                _d(np.ascontiguousarray(pb.plane_obs_z[b:e])), _d(pb.plane_normal[f]), float(pb.plane_distance[f]), float(pb.plane_sqrt_inv_cov), _d(r), None)
            s = float(r @ r)
            out.append(r * (np.sqrt(np.log1p(s) / s) if s > 0 else 1.0))
        return np.concatenate(out)


@pytest.mark.parametrize("kw,slack", [(dict(n_frames=4, n_landmarks=30), 2e-6), (dict(n_frames=5, n_landmarks=40, plane_fraction=0.5), 2e-5)], ids=["vision", "plane"])
def test_oracle_solution_is_a_minimum_of_the_stated_objective_for_scipy(oracle, kw, slack):
    """Vision-only windows (frame 0 fixed; with and without plane-distance factors): one scalar objective, no live-bias quirk.
    The oracle runs to convergence; scipy's trust-region-reflective least squares (numerical Jacobian, its own step control),
    started from the oracle's solution, (a) evaluates the objective to the cost the oracle reports and (b) cannot lower it by
    more than the function tolerance the oracle stopped at.  (No bound on how far scipy moves: with only frame 0 fixed the
    monocular scale is a flat direction of this objective, and scipy drifts along it.  The plane window gets ten times the slack:
    the plane-distance factor's analytic Jacobian carries the reference's sign quirk (oracle_factors.h), so the oracle's fixed
    point is not exactly a stationary point of the objective there -- scipy finds 6e-6 of relative decrease in 40 evaluations.)"""
    pb = synth.make_window(**kw)
    pb.max_iterations = 200
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)
    assert sm.termination == 0  # CONVERGENCE
    obj = _Objective(pb, oracle)
    obj.fs0, obj.rho0 = st.frame_state.copy(), st.lm_inv_depth.copy()
    n = obj.D.ncols
    f0 = obj.rows(np.zeros(n))
    cost0 = 0.5 * float(f0 @ f0)
    np.testing.assert_allclose(cost0, sm.final_cost, rtol=1e-12)
    res = optimize.least_squares(obj.rows, np.zeros(n), method="trf", jac="2-point", x_scale=1.0, xtol=1e-13, ftol=1e-13, gtol=1e-11, max_nfev=40)
    assert res.cost <= cost0 * (1 + 1e-12)
    assert cost0 - res.cost <= slack * cost0, (cost0, res.cost)  # Ceres' function_tolerance (1e-6) is where the oracle stops
