"""ctypes mirror of include/pvio_hip.h and the loader for the product library.

The product library is `pvio_amd/lib/libpvio_hip.so` (hand-written HIP for gfx950 + host C++).  There is
NO CPU fallback: if the library is missing, `load()` raises, and every device entry point returns a
negative status when no GPU is present.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libpvio_hip.so")

PVIO_OK = 0
TERM_CONVERGENCE, TERM_NO_CONVERGENCE, TERM_FAILURE = 0, 1, 2

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
c_float_p = C.POINTER(C.c_float)
c_int16_p = C.POINTER(C.c_int16)


class HipOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32), ("use_graph", C.c_int32),
                ("debug_fail_factorizations", C.c_int32), ("debug_invalid_steps", C.c_int32), ("linearize_mode", C.c_int32), ("debug_force_sharded", C.c_int32), ("reuse_identical_candidates", C.c_int32)]


class BAProblemC(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32), ("n_landmarks", C.c_int32), ("n_obs", C.c_int32), ("use_inertial", C.c_int32),
        ("frame_fixed", c_uint8_p), ("cam_extrinsic", c_double_p), ("imu_extrinsic", c_double_p),
        ("sqrt_inv_cov", c_double_p), ("intrinsics", c_double_p),
        ("lm_anchor_frame", c_int32_p), ("lm_anchor_z", c_double_p), ("lm_obs_ptr", c_int32_p),
        ("obs_frame", c_int32_p), ("obs_z", c_double_p),
        ("preint_valid", c_uint8_p), ("preint_delta", c_double_p), ("preint_sqrt_inv_cov", c_double_p),
        ("preint_jacobian", c_double_p),
        ("prior_n", C.c_int32), ("n_plane_factors", C.c_int32),
        ("prior_frames", c_int32_p), ("prior_S", c_double_p), ("prior_s", c_double_p), ("prior_lin_state", c_double_p),
        ("plane_obs_ptr", c_int32_p), ("plane_obs_frame", c_int32_p), ("plane_obs_z", c_double_p),
        ("plane_normal", c_double_p), ("plane_distance", c_double_p), ("plane_sqrt_inv_cov", C.c_double),
        ("max_iterations", C.c_int32), ("reserved0", C.c_int32), ("max_solver_time", C.c_double),
        ("n_rot_priors", C.c_int32), ("reserved1", C.c_int32),
        ("rot_prior_frame", c_int32_p), ("rot_prior_q0", c_double_p), ("rot_prior_sqrt_info", c_double_p),
        ("lm_multiplicity", c_int32_p),
    ]


class BAStateC(C.Structure):
    _fields_ = [("frame_state", c_double_p), ("lm_inv_depth", c_double_p), ("lm_quality", c_double_p),
                ("lm_valid", c_uint8_p)]


class BAIterationC(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32),
                ("reserved", C.c_int32), ("cost", C.c_double), ("cost_change", C.c_double),
                ("gradient_max_norm", C.c_double), ("step_norm", C.c_double), ("relative_decrease", C.c_double),
                ("trust_region_radius", C.c_double), ("mu", C.c_double)]


class BASummaryC(C.Structure):
    _fields_ = [("termination", C.c_int32), ("is_usable", C.c_int32), ("num_iterations", C.c_int32),
                ("num_successful_steps", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("solve_seconds", C.c_double), ("device_seconds", C.c_double),
                ("trace_capacity", C.c_int32), ("trace_len", C.c_int32),
                ("trace", C.POINTER(BAIterationC)), ("trace_states", c_double_p)]


class BAPriorC(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("S", c_double_p), ("s", c_double_p),
                ("info_matrix", c_double_p), ("info_vector", c_double_p)]


class BAKernelTimesC(C.Structure):
    _fields_ = [("total_ms", C.c_double * 4), ("launches", C.c_int32 * 4), ("phase_ticks", (C.c_int64 * 32) * 4),
                ("comm_ms", C.c_double * 2), ("comm_launches", C.c_int32 * 2)]


class ImuNoiseC(C.Structure):
    _fields_ = [("cov_w", C.c_double * 9), ("cov_a", C.c_double * 9), ("cov_bg", C.c_double * 9),
                ("cov_ba", C.c_double * 9)]


ABI_VERSION = 2  # PVIO_HIP_ABI_VERSION of include/pvio_hip.h these ctypes structs mirror

# every symbol include/pvio_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "pvio_hip_create", "pvio_hip_destroy", "pvio_hip_last_error", "pvio_hip_version", "pvio_hip_abi_version",
    "pvio_hip_ba_solve", "pvio_hip_ba_marginalize", "pvio_hip_ba_reprojection_error",
    "pvio_hip_ba_upload", "pvio_hip_ba_solve_resident", "pvio_hip_ba_download", "pvio_hip_ba_profile_resident", "pvio_hip_ba_last_candidate_repeats", "pvio_hip_ba_graph_replays",
    "pvio_hip_comm_unique_id", "pvio_hip_comm_init", "pvio_preintegrate",
    "pvio_hip_image_create", "pvio_hip_image_release", "pvio_hip_image_download_level", "pvio_hip_klt_track", "pvio_hip_image_detect", "pvio_hip_image_download_response",
    "pvio_hip_klt_last_device_ms", "pvio_hip_fundamental_ransac", "pvio_hip_ransac_last_hypotheses",
    "pvio_hip_undistort_create", "pvio_hip_undistort_release", "pvio_hip_image_create_undistorted",
]

_lib = None


def load(path=None):
    """Load libpvio_hip.so.  Raises if it has not been built -- the product has no fallback path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PVIO_HIP_LIB") or LIB_PATH  # (PVIO_HIP_LIB: another BUILD of the same library, for same-box experiments)
    if not os.path.exists(p):
        raise RuntimeError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.pvio_hip_create.argtypes = [C.POINTER(HipOpts), C.POINTER(vp)]
    lib.pvio_hip_create.restype = C.c_int32
    lib.pvio_hip_destroy.argtypes = [vp]
    lib.pvio_hip_destroy.restype = None
    lib.pvio_hip_last_error.argtypes = [vp]
    lib.pvio_hip_last_error.restype = C.c_char_p
    lib.pvio_hip_version.argtypes = []
    lib.pvio_hip_version.restype = C.c_char_p
    lib.pvio_hip_abi_version.argtypes = []
    lib.pvio_hip_abi_version.restype = C.c_int32
    if lib.pvio_hip_abi_version() != ABI_VERSION:  # the ctypes mirrors below are a layout of their own: same check as a C caller makes
        raise RuntimeError("%s has ABI version %d, pvio_amd/capi.py mirrors version %d of include/pvio_hip.h" % (p, lib.pvio_hip_abi_version(), ABI_VERSION))
    lib.pvio_hip_ba_solve.argtypes = [vp, C.POINTER(BAProblemC), C.POINTER(BAStateC), C.POINTER(BASummaryC)]
    lib.pvio_hip_ba_solve.restype = C.c_int32
    lib.pvio_hip_ba_marginalize.argtypes = [vp, C.POINTER(BAProblemC), C.POINTER(BAStateC), C.c_int32, C.POINTER(BAPriorC)]
    lib.pvio_hip_ba_marginalize.restype = C.c_int32
    lib.pvio_hip_ba_reprojection_error.argtypes = [vp, C.POINTER(BAProblemC), C.POINTER(BAStateC), c_double_p]
    lib.pvio_hip_ba_reprojection_error.restype = C.c_int32
    lib.pvio_hip_ba_upload.argtypes = [vp, C.POINTER(BAProblemC), C.POINTER(BAStateC)]
    lib.pvio_hip_ba_upload.restype = C.c_int32
    lib.pvio_hip_ba_solve_resident.argtypes = [vp, C.POINTER(BASummaryC)]
    lib.pvio_hip_ba_solve_resident.restype = C.c_int32
    lib.pvio_hip_ba_profile_resident.argtypes = [vp, C.POINTER(BASummaryC), C.POINTER(BAKernelTimesC)]
    lib.pvio_hip_ba_profile_resident.restype = C.c_int32
    lib.pvio_hip_ba_download.argtypes = [vp, C.POINTER(BAStateC)]
    lib.pvio_hip_ba_download.restype = C.c_int32
    lib.pvio_hip_comm_unique_id.argtypes = [c_uint8_p]
    lib.pvio_hip_comm_unique_id.restype = C.c_int32
    lib.pvio_hip_comm_init.argtypes = [vp, c_uint8_p, C.c_int32, C.c_int32]
    lib.pvio_hip_comm_init.restype = C.c_int32
    lib.pvio_preintegrate.argtypes = [C.c_int32, c_double_p, c_double_p, c_double_p, C.c_double, c_double_p, c_double_p,
                                      C.POINTER(ImuNoiseC), c_double_p, c_double_p, c_double_p, c_double_p]
    lib.pvio_preintegrate.restype = C.c_int32
    lib.pvio_hip_image_create.argtypes = [vp, c_uint8_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.pvio_hip_image_create.restype = C.c_int32
    c_uint16_p = C.POINTER(C.c_uint16)
    lib.pvio_hip_undistort_create.argtypes = [vp, c_int16_p, c_uint16_p, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.pvio_hip_undistort_create.restype = C.c_int32
    lib.pvio_hip_undistort_release.argtypes = [vp, vp]
    lib.pvio_hip_undistort_release.restype = None
    lib.pvio_hip_image_create_undistorted.argtypes = [vp, vp, c_uint8_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.pvio_hip_image_create_undistorted.restype = C.c_int32
    lib.pvio_hip_image_release.argtypes = [vp, vp]
    lib.pvio_hip_image_release.restype = None
    lib.pvio_hip_image_download_level.argtypes = [vp, vp, C.c_int32, c_uint8_p, c_int16_p, c_int32_p, c_int32_p]
    lib.pvio_hip_image_download_level.restype = C.c_int32
    lib.pvio_hip_klt_track.argtypes = [vp, vp, vp, C.c_int32, c_float_p, c_float_p, c_uint8_p]
    lib.pvio_hip_klt_track.restype = C.c_int32
    lib.pvio_hip_image_detect.argtypes = [vp, vp, C.c_int32, C.c_double, C.c_double, c_float_p, c_float_p, C.POINTER(C.c_int32)]
    lib.pvio_hip_image_detect.restype = C.c_int32
    lib.pvio_hip_image_download_response.argtypes = [vp, vp, c_float_p]
    lib.pvio_hip_image_download_response.restype = C.c_int32
    lib.pvio_hip_klt_last_device_ms.argtypes = [vp]
    lib.pvio_hip_klt_last_device_ms.restype = C.c_double
    lib.pvio_hip_fundamental_ransac.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_double, C.c_double, C.c_int32, c_uint8_p, c_double_p,
                                                C.POINTER(C.c_int32)]
    lib.pvio_hip_fundamental_ransac.restype = C.c_int32
    lib.pvio_hip_ransac_last_hypotheses.argtypes = [vp]
    lib.pvio_hip_ransac_last_hypotheses.restype = C.c_int32
    if path is None:
        _lib = lib
    return lib
