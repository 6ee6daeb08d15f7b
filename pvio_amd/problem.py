"""Flat SoA bundle-adjustment window (numpy) <-> `pvio_ba_problem` (include/pvio_hip.h).

This is the Python mirror of what the C++ adapter (pvio_amd/host/bundle_adjustor.cpp) builds from a
pvio::Map in the reference's residual-block order (bundle_adjustor.cpp:75-242).
"""
import ctypes as C

import numpy as np

from . import capi


def _p(arr, typ):
    if arr is None:
        return C.cast(None, typ)
    return arr.ctypes.data_as(typ)


class BAProblem:
    """All arrays are C-contiguous numpy; shapes follow include/pvio_hip.h."""

    def __init__(self, n_frames):
        N = n_frames
        self.use_inertial = False
        self.frame_fixed = np.zeros(N, np.uint8)
        self.cam_extrinsic = np.zeros((N, 7))
        self.imu_extrinsic = np.zeros((N, 7))
        self.cam_extrinsic[:, 3] = 1.0
        self.imu_extrinsic[:, 3] = 1.0
        self.sqrt_inv_cov = np.zeros((N, 4))
        self.intrinsics = np.zeros((N, 4))
        self.lm_anchor_frame = np.zeros(0, np.int32)
        self.lm_anchor_z = np.zeros((0, 2))
        self.lm_obs_ptr = np.zeros(1, np.int32)
        self.obs_frame = np.zeros(0, np.int32)
        self.obs_z = np.zeros((0, 2))
        self.preint_valid = np.zeros(N, np.uint8)
        self.preint_delta = np.zeros((N, 11))
        self.preint_delta[:, 4] = 1.0
        self.preint_sqrt_inv_cov = np.zeros((N, 225))
        self.preint_jacobian = np.zeros((N, 45))
        self.prior_frames = np.zeros(0, np.int32)
        self.prior_S = np.zeros((0, 0))
        self.prior_s = np.zeros(0)
        self.prior_lin_state = np.zeros((0, 16))
        self.plane_obs_ptr = np.zeros(1, np.int32)
        self.plane_obs_frame = np.zeros(0, np.int32)
        self.plane_obs_z = np.zeros((0, 2))
        self.plane_normal = np.zeros((0, 3))
        self.plane_distance = np.zeros(0)
        self.plane_sqrt_inv_cov = 0.0
        # rotation priors (RotationPriorFactor; no reference counterpart, include/pvio_hip.h)
        self.rot_prior_frame = np.zeros(0, np.int32)
        self.rot_prior_q0 = np.zeros((0, 4))
        self.rot_prior_sqrt_info = np.zeros((0, 9))
        # duplicate residual blocks (bundle_adjustor.cpp:165-179): how often each landmark's blocks are listed; None = once
        self.lm_multiplicity = None
        self.max_iterations = 10
        self.max_solver_time = 1.0e6
        # states (initial guess) and, for synthetic windows, the ground truth
        self.frame_state = np.zeros((N, 16))
        self.frame_state[:, 3] = 1.0
        self.lm_inv_depth = np.zeros(0)
        self.truth_frame_state = None
        self.truth_inv_depth = None
        self.meta = {}

    @property
    def n_frames(self):
        return self.frame_fixed.shape[0]

    @property
    def n_landmarks(self):
        return self.lm_anchor_frame.shape[0]

    @property
    def n_obs(self):
        return self.obs_frame.shape[0]

    @property
    def n_plane_factors(self):
        return self.plane_normal.shape[0]

    def state_dim(self):
        return self.n_frames * 16 + self.n_landmarks

    def _canon(self):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        for name in ["cam_extrinsic", "imu_extrinsic", "sqrt_inv_cov", "intrinsics", "lm_anchor_z", "obs_z", "preint_delta",
                     "preint_sqrt_inv_cov", "preint_jacobian", "prior_S", "prior_s", "prior_lin_state", "plane_obs_z",
                     "plane_normal", "plane_distance", "frame_state", "lm_inv_depth", "rot_prior_q0", "rot_prior_sqrt_info"]:
            setattr(self, name, f64(getattr(self, name)))
        for name in ["lm_anchor_frame", "lm_obs_ptr", "obs_frame", "prior_frames", "plane_obs_ptr", "plane_obs_frame", "rot_prior_frame"]:
            setattr(self, name, i32(getattr(self, name)))
        for name in ["frame_fixed", "preint_valid"]:
            setattr(self, name, u8(getattr(self, name)))
        if self.lm_multiplicity is not None:
            self.lm_multiplicity = i32(self.lm_multiplicity)

    def as_c(self):
        """Returns a `pvio_ba_problem` whose pointers alias this object's arrays (keep `self` alive)."""
        self._canon()
        pb = capi.BAProblemC()
        pb.n_frames, pb.n_landmarks, pb.n_obs = self.n_frames, self.n_landmarks, self.n_obs
        pb.use_inertial = int(self.use_inertial)
        dp, ip, bp = capi.c_double_p, capi.c_int32_p, capi.c_uint8_p
        pb.frame_fixed = _p(self.frame_fixed, bp)
        pb.cam_extrinsic = _p(self.cam_extrinsic, dp)
        pb.imu_extrinsic = _p(self.imu_extrinsic, dp)
        pb.sqrt_inv_cov = _p(self.sqrt_inv_cov, dp)
        pb.intrinsics = _p(self.intrinsics, dp)
        pb.lm_anchor_frame = _p(self.lm_anchor_frame, ip)
        pb.lm_anchor_z = _p(self.lm_anchor_z, dp)
        pb.lm_obs_ptr = _p(self.lm_obs_ptr, ip)
        pb.obs_frame = _p(self.obs_frame, ip)
        pb.obs_z = _p(self.obs_z, dp)
        pb.preint_valid = _p(self.preint_valid, bp)
        pb.preint_delta = _p(self.preint_delta, dp)
        pb.preint_sqrt_inv_cov = _p(self.preint_sqrt_inv_cov, dp)
        pb.preint_jacobian = _p(self.preint_jacobian, dp)
        pb.prior_n = int(self.prior_frames.shape[0])
        pb.n_plane_factors = self.n_plane_factors
        pb.prior_frames = _p(self.prior_frames, ip)
        pb.prior_S = _p(self.prior_S, dp)
        pb.prior_s = _p(self.prior_s, dp)
        pb.prior_lin_state = _p(self.prior_lin_state, dp)
        pb.plane_obs_ptr = _p(self.plane_obs_ptr, ip)
        pb.plane_obs_frame = _p(self.plane_obs_frame, ip)
        pb.plane_obs_z = _p(self.plane_obs_z, dp)
        pb.plane_normal = _p(self.plane_normal, dp)
        pb.plane_distance = _p(self.plane_distance, dp)
        pb.plane_sqrt_inv_cov = float(self.plane_sqrt_inv_cov)
        pb.max_iterations = int(self.max_iterations)
        pb.max_solver_time = float(self.max_solver_time)
        pb.n_rot_priors = int(self.rot_prior_frame.shape[0])
        pb.rot_prior_frame = _p(self.rot_prior_frame, ip)
        pb.rot_prior_q0 = _p(self.rot_prior_q0, dp)
        pb.rot_prior_sqrt_info = _p(self.rot_prior_sqrt_info, dp)
        pb.lm_multiplicity = _p(self.lm_multiplicity, ip)
        return pb

    def shard(self, rank, world):
        """Landmark shard `rank` of `world` (contiguous CSR ranges balanced on factor count); frames,
        pre-integration, prior are replicated; plane factors are split the same way."""
        if world == 1:
            return self
        self._canon()
        import copy
        out = copy.copy(self)
        F, M = self.n_obs, self.n_landmarks
        ptr = self.lm_obs_ptr.astype(np.int64)
        targets = [(F * r) // world for r in range(world + 1)]
        cuts = [int(np.searchsorted(ptr, t, side="left")) for t in targets]
        cuts[0], cuts[-1] = 0, M
        l0, l1 = cuts[rank], cuts[rank + 1]
        o0, o1 = int(ptr[l0]), int(ptr[l1])
        out.lm_anchor_frame = self.lm_anchor_frame[l0:l1].copy()
        out.lm_anchor_z = self.lm_anchor_z[l0:l1].copy()
        out.lm_obs_ptr = (self.lm_obs_ptr[l0:l1 + 1] - o0).astype(np.int32)
        out.obs_frame = self.obs_frame[o0:o1].copy()
        out.obs_z = self.obs_z[o0:o1].copy()
        out.lm_inv_depth = self.lm_inv_depth[l0:l1].copy()
        if self.lm_multiplicity is not None:
            out.lm_multiplicity = self.lm_multiplicity[l0:l1].copy()
        if self.truth_inv_depth is not None:
            out.truth_inv_depth = self.truth_inv_depth[l0:l1].copy()
        Pn = self.n_plane_factors
        p0, p1 = (Pn * rank) // world, (Pn * (rank + 1)) // world
        q0, q1 = int(self.plane_obs_ptr[p0]), int(self.plane_obs_ptr[p1])
        out.plane_obs_ptr = (self.plane_obs_ptr[p0:p1 + 1] - q0).astype(np.int32)
        out.plane_obs_frame = self.plane_obs_frame[q0:q1].copy()
        out.plane_obs_z = self.plane_obs_z[q0:q1].copy()
        out.plane_normal = self.plane_normal[p0:p1].copy()
        out.plane_distance = self.plane_distance[p0:p1].copy()
        out.frame_state = self.frame_state.copy()
        out.meta = dict(self.meta, shard=(rank, world), lm_range=(l0, l1))
        return out


class BAState:
    def __init__(self, problem):
        self.frame_state = np.array(problem.frame_state, dtype=np.float64, order="C", copy=True)
        self.lm_inv_depth = np.array(problem.lm_inv_depth, dtype=np.float64, order="C", copy=True)
        self.lm_quality = np.full(problem.n_landmarks, -1.0)
        self.lm_valid = np.ones(problem.n_landmarks, np.uint8)

    def as_c(self):
        st = capi.BAStateC()
        st.frame_state = _p(self.frame_state, capi.c_double_p)
        st.lm_inv_depth = _p(self.lm_inv_depth, capi.c_double_p)
        st.lm_quality = _p(self.lm_quality, capi.c_double_p)
        st.lm_valid = _p(self.lm_valid, capi.c_uint8_p)
        return st

    def vector(self):
        return np.concatenate([self.frame_state.ravel(), self.lm_inv_depth])


class BASummary:
    def __init__(self, problem, trace=True):
        cap = problem.max_iterations + 2 if trace else 0
        self.cap = cap
        self._trace = (capi.BAIterationC * max(cap, 1))()
        self.trace_states = np.zeros((max(cap, 1), problem.state_dim())) if trace else None
        self.c = capi.BASummaryC()
        self.c.trace_capacity = cap
        self.c.trace = C.cast(self._trace, C.POINTER(capi.BAIterationC)) if trace else None
        self.c.trace_states = _p(self.trace_states, capi.c_double_p) if trace else None

    def __getattr__(self, k):
        if k in ("termination", "is_usable", "num_iterations", "num_successful_steps", "initial_cost", "final_cost",
                 "solve_seconds", "device_seconds", "trace_len"):
            return getattr(self.c, k)
        raise AttributeError(k)

    def trace(self):
        out = []
        for i in range(self.c.trace_len):
            t = self._trace[i]
            out.append({f[0]: getattr(t, f[0]) for f in capi.BAIterationC._fields_ if f[0] != "reserved"})
        return out
