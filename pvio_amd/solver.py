"""Thin ctypes driver over the C-ABI (include/pvio_hip.h) used by tests and bench.py.

It adds nothing to the product path: every call goes straight into libpvio_hip.so.
"""
import ctypes as C

import numpy as np

from . import capi
from .problem import BAState, BASummary


class HipError(RuntimeError):
    pass


class HipContext:
    def __init__(self, device=0, rank=0, world_size=1, use_graph=True, lib=None, debug_fail_factorizations=0, debug_invalid_steps=0,
                 linearize_mode=0, force_sharded=False, reuse_identical_candidates=False):
        self.lib = lib or capi.load()
        opts = capi.HipOpts()
        opts.device, opts.rank, opts.world_size, opts.use_graph = device, rank, world_size, int(use_graph)
        opts.debug_fail_factorizations, opts.debug_invalid_steps = debug_fail_factorizations, debug_invalid_steps  # tests only
        opts.linearize_mode = linearize_mode
        opts.debug_force_sharded = int(force_sharded)
        opts.reuse_identical_candidates = int(reuse_identical_candidates)  # pvio_hip_opts: skip re-evaluating a bit-identical rejected candidate
        self.ctx = C.c_void_p()
        rc = self.lib.pvio_hip_create(C.byref(opts), C.byref(self.ctx))
        if rc != 0:
            raise HipError("pvio_hip_create failed with status %d (no usable GPU? there is no CPU fallback)" % rc)
        self._keep = None

    def close(self):
        if self.ctx:
            self.lib.pvio_hip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.pvio_hip_last_error(self.ctx)
            raise HipError("%s failed: status %d (%s)" % (what, rc, msg.decode() if msg else ""))

    # one-shot: upload + solve + download (what the BundleAdjustor adapter calls)
    def solve(self, problem, state=None, summary=None, trace=True):
        state = state or BAState(problem)
        summary = summary or BASummary(problem, trace=trace)
        pb, st = problem.as_c(), state.as_c()
        self._check(self.lib.pvio_hip_ba_solve(self.ctx, C.byref(pb), C.byref(st), C.byref(summary.c)), "pvio_hip_ba_solve")
        return state, summary

    # device-resident variant (bench): upload once, solve many times from the same initial state
    def upload(self, problem, state=None):
        state = state or BAState(problem)
        pb, st = problem.as_c(), state.as_c()
        self._keep = (problem, state)
        self._check(self.lib.pvio_hip_ba_upload(self.ctx, C.byref(pb), C.byref(st)), "pvio_hip_ba_upload")
        return state

    def graph_replays(self):
        """solves of this context that ran as a replay of the captured slot graph"""
        self.lib.pvio_hip_ba_graph_replays.argtypes = [C.c_void_p]
        return int(self.lib.pvio_hip_ba_graph_replays(self.ctx))

    def last_candidate_repeats(self):
        """candidate evaluations the last solve short-circuited (reuse_identical_candidates)"""
        self.lib.pvio_hip_ba_last_candidate_repeats.argtypes = [C.c_void_p]
        return int(self.lib.pvio_hip_ba_last_candidate_repeats(self.ctx))

    def solve_resident(self, summary):
        self._check(self.lib.pvio_hip_ba_solve_resident(self.ctx, C.byref(summary.c)), "pvio_hip_ba_solve_resident")
        return summary

    def profile_resident(self, summary):
        """One resident solve with hipEvents around every kernel launch -> {kernel: (total_ms, launches)}."""
        kt = capi.BAKernelTimesC()
        self._check(self.lib.pvio_hip_ba_profile_resident(self.ctx, C.byref(summary.c), C.byref(kt)), "pvio_hip_ba_profile_resident")
        names = ["k_linearize", "k_reduce", "k_dense", "k_backsub"]
        self.last_phase_ticks = {n: [int(x) for x in kt.phase_ticks[i]] for i, n in enumerate(names)}
        # landmark-sharded solves: the two exchange steps of an iteration (zero on a single GPU)
        self.last_comm = {"allreduce_system": (kt.comm_ms[0], kt.comm_launches[0]), "allreduce_backsub": (kt.comm_ms[1], kt.comm_launches[1])}
        return {n: (kt.total_ms[i], kt.launches[i]) for i, n in enumerate(names)}

    def download(self, state):
        st = state.as_c()
        self._check(self.lib.pvio_hip_ba_download(self.ctx, C.byref(st)), "pvio_hip_ba_download")
        return state

    def reprojection_error(self, problem, state):
        pb, st = problem.as_c(), state.as_c()
        out = C.c_double(0)
        self._check(self.lib.pvio_hip_ba_reprojection_error(self.ctx, C.byref(pb), C.byref(st), C.byref(out)), "reprojection_error")
        return out.value

    def marginalize(self, problem, state, victim, want_info=True):
        pb, st = problem.as_c(), state.as_c()
        n = problem.n_frames - 1
        S, s = np.zeros((15 * n, 15 * n)), np.zeros(15 * n)
        IM, iv = np.zeros((15 * n, 15 * n)), np.zeros(15 * n)
        pr = capi.BAPriorC()
        pr.S, pr.s = S.ctypes.data_as(capi.c_double_p), s.ctypes.data_as(capi.c_double_p)
        if want_info:
            pr.info_matrix, pr.info_vector = IM.ctypes.data_as(capi.c_double_p), iv.ctypes.data_as(capi.c_double_p)
        self._check(self.lib.pvio_hip_ba_marginalize(self.ctx, C.byref(pb), C.byref(st), int(victim), C.byref(pr)), "pvio_hip_ba_marginalize")
        return S, s, IM, iv


class HipUndistort:
    """Device-resident fixed-point remap tables (pvio_hip_undistort_create): map_xy int16 [h][w][2], map_frac uint16 [h][w]."""

    def __init__(self, ctx, map_xy, map_frac):
        self.ctx = ctx
        xy = np.ascontiguousarray(map_xy, dtype=np.int16)
        fr = np.ascontiguousarray(map_frac, dtype=np.uint16)
        self.h, self.w = fr.shape
        assert xy.shape == (self.h, self.w, 2)
        self.handle = C.c_void_p()
        ctx._check(ctx.lib.pvio_hip_undistort_create(ctx.ctx, xy.ctypes.data_as(capi.c_int16_p), fr.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                     self.w, self.h, C.byref(self.handle)), "pvio_hip_undistort_create")

    def release(self):
        if self.handle:
            self.ctx.lib.pvio_hip_undistort_release(self.ctx.ctx, self.handle)
            self.handle = C.c_void_p()


class HipImage:
    """Device-resident CLAHE'd pyramid (pvio_hip_image_create); mirrors OpenCvImage::preprocess.  With `undistort` the
    pixels are the distorted camera image and are remapped on the device first (the dataset readers' cv::undistort / remap)."""

    def __init__(self, ctx, pixels, clahe=True, undistort=None):
        self.ctx = ctx
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        self.h, self.w = px.shape
        self.handle = C.c_void_p()
        if undistort is not None:
            ctx._check(ctx.lib.pvio_hip_image_create_undistorted(ctx.ctx, undistort.handle, px.ctypes.data_as(capi.c_uint8_p), self.w, self.h, self.w,
                                                                 int(clahe), C.byref(self.handle)), "pvio_hip_image_create_undistorted")
            self.h, self.w = undistort.h, undistort.w
            return
        ctx._check(ctx.lib.pvio_hip_image_create(ctx.ctx, px.ctypes.data_as(capi.c_uint8_p), self.w, self.h, self.w, int(clahe), C.byref(self.handle)),
                   "pvio_hip_image_create")

    def level(self, l):
        w, h = C.c_int32(0), C.c_int32(0)
        self.ctx._check(self.ctx.lib.pvio_hip_image_download_level(self.ctx.ctx, self.handle, l, None, None, C.byref(w), C.byref(h)), "download_level")
        img = np.zeros((h.value, w.value), np.uint8)
        drv = np.zeros((h.value, w.value, 2), np.int16)
        self.ctx._check(self.ctx.lib.pvio_hip_image_download_level(self.ctx.ctx, self.handle, l, img.ctypes.data_as(capi.c_uint8_p),
                                                                 drv.ctypes.data_as(capi.c_int16_p), C.byref(w), C.byref(h)), "download_level")
        return img, drv

    def release(self):
        if self.handle:
            self.ctx.lib.pvio_hip_image_release(self.ctx.ctx, self.handle)
            self.handle = C.c_void_p()


def klt_track(ctx, prev, nxt, prev_xy, next_xy_init):
    """pvio_hip_klt_track: returns (next_xy, status, device_ms); mirrors OpenCvImage::track_keypoints up to the border kill."""
    p = np.ascontiguousarray(prev_xy, dtype=np.float32)
    q = np.array(next_xy_init, dtype=np.float32, order="C", copy=True)
    st = np.zeros(p.shape[0], np.uint8)
    ctx._check(ctx.lib.pvio_hip_klt_track(ctx.ctx, prev.handle, nxt.handle, p.shape[0], p.ctypes.data_as(capi.c_float_p), q.ctypes.data_as(capi.c_float_p),
                                          st.ctypes.data_as(capi.c_uint8_p)), "pvio_hip_klt_track")
    return q, st, ctx.lib.pvio_hip_klt_last_device_ms(ctx.ctx)


def fundamental_ransac(ctx, p_xy, q_xy, threshold=1.0, confidence=0.99, max_iterations=1000):
    """pvio_hip_fundamental_ransac: cv::findFundamentalMat(FM_RANSAC) with the hypotheses evaluated in batches on the device.
    -> (inlier count, mask uint8 [n], F 3x3, hypotheses evaluated)"""
    p = np.ascontiguousarray(p_xy, dtype=np.float32)
    q = np.ascontiguousarray(q_xy, dtype=np.float32)
    n = p.shape[0]
    mask, F, good = np.zeros(max(n, 1), np.uint8), np.zeros(9), C.c_int32(0)
    fp = C.POINTER(C.c_float)
    ctx._check(ctx.lib.pvio_hip_fundamental_ransac(ctx.ctx, n, p.ctypes.data_as(fp), q.ctypes.data_as(fp), float(threshold), float(confidence), int(max_iterations),
                                                   mask.ctypes.data_as(capi.c_uint8_p), F.ctypes.data_as(capi.c_double_p), C.byref(good)), "fundamental_ransac")
    return good.value, mask[:n], F.reshape(3, 3), int(ctx.lib.pvio_hip_ransac_last_hypotheses(ctx.ctx))


def preintegrate(t, w, a, t_end, bg, ba, noise, lib=None):
    """Product host-side pre-integration (pvio_preintegrate); same signature as oracle_py.preintegrate."""
    lib = lib or capi.load()
    f64 = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    t, w, a, bg, ba = f64(t), f64(w), f64(a), f64(bg), f64(ba)
    delta, cov, U, jac = np.zeros(11), np.zeros(225), np.zeros(225), np.zeros(45)
    nz = capi.ImuNoiseC()
    for k in ("cov_w", "cov_a", "cov_bg", "cov_ba"):
        getattr(nz, k)[:] = list(np.asarray(noise[k], float).ravel())
    p = lambda x: x.ctypes.data_as(capi.c_double_p)
    rc = lib.pvio_preintegrate(len(t), p(t), p(w), p(a), float(t_end), p(bg), p(ba), C.byref(nz), p(delta), p(cov), p(U), p(jac))
    if rc != 0:
        raise HipError("pvio_preintegrate failed: %d" % rc)
    return delta, cov, U, jac


def detect_corners(ctx, img, max_corners=1000, quality=1.0e-3, min_distance=20.0, want_response_map=False):
    """pvio_hip_image_detect: Harris corners (goodFeaturesToTrack semantics) of a preprocessed HipImage.
    Returns (xy float32 [n, 2], response float32 [n]) and, on request, the response map."""
    xy = np.zeros((max_corners, 2), np.float32)
    resp = np.zeros(max_corners, np.float32)
    n = C.c_int32(0)
    ctx._check(ctx.lib.pvio_hip_image_detect(ctx.ctx, img.handle, max_corners, quality, min_distance, xy.ctypes.data_as(capi.c_float_p),
                                             resp.ctypes.data_as(capi.c_float_p), C.byref(n)), "pvio_hip_image_detect")
    if not want_response_map:
        return xy[:n.value].copy(), resp[:n.value].copy()
    rmap = np.zeros((img.h, img.w), np.float32)
    ctx._check(ctx.lib.pvio_hip_image_download_response(ctx.ctx, img.handle, rmap.ctypes.data_as(capi.c_float_p)), "pvio_hip_image_download_response")
    return xy[:n.value].copy(), resp[:n.value].copy(), rmap
