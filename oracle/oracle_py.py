"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (pvio_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pvio_amd import capi  # noqa: E402  (struct layouts only)

LIB = os.path.join(HERE, "liboracle.so")
dp = capi.c_double_p


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".cpp", ".h"))]
    srcs.append(os.path.join(HERE, "..", "include", "pvio_hip.h"))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    return LIB


def solve_fast(problem, state, summary):
    """oracle_ba_solve of the contraction-on build (liboracle_fast.so): bench.py's timed cpu_baseline only"""
    global _fast
    if _fast is None:
        # rebuilt on the box it is timed on: -march=native must mean THIS host
        subprocess.check_call(["make", "-s", "-B", "-C", HERE, "liboracle_fast.so"])
        _fast = C.CDLL(os.path.join(HERE, "liboracle_fast.so"))
        _fast.oracle_ba_solve.argtypes = [C.POINTER(capi.BAProblemC), C.POINTER(capi.BAStateC), C.POINTER(capi.BASummaryC)]
    pbc, stc = problem.as_c(), state.as_c()
    rc = _fast.oracle_ba_solve(C.byref(pbc), C.byref(stc), C.byref(summary.c))
    assert rc == 0
    return state, summary


_lib = None
_fast = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        L = _lib
        L.oracle_ba_solve.argtypes = [C.POINTER(capi.BAProblemC), C.POINTER(capi.BAStateC), C.POINTER(capi.BASummaryC)]
        L.oracle_ba_linearize.argtypes = [C.POINTER(capi.BAProblemC), dp, dp, dp, dp, dp, dp, dp, dp, capi.c_int32_p, capi.c_int32_p]
        L.oracle_ba_cost.argtypes = [C.POINTER(capi.BAProblemC), dp, dp, dp, dp]
        L.oracle_ba_reprojection_error.argtypes = [C.POINTER(capi.BAProblemC), C.POINTER(capi.BAStateC), dp]
        L.oracle_ba_marginalize.argtypes = [C.POINTER(capi.BAProblemC), C.POINTER(capi.BAStateC), C.c_int32, C.POINTER(capi.BAPriorC)]
        L.oracle_preintegrate.argtypes = [C.c_int32, dp, dp, dp, C.c_double, dp, dp, C.POINTER(capi.ImuNoiseC), dp, dp, dp, dp]
        L.oracle_eval_reprojection.argtypes = [dp, dp, C.c_double, dp, dp, dp, dp, dp, dp, dp]
        L.oracle_eval_reprojection.restype = None
        L.oracle_eval_preintegration.argtypes = [dp] * 10
        L.oracle_eval_preintegration.restype = None
        L.oracle_eval_prior.argtypes = [C.c_int32, dp, dp, dp, dp, dp, dp]
        L.oracle_eval_prior.restype = None
        L.oracle_eval_plane.argtypes = [C.c_int32, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp]
        L.oracle_eval_plane.restype = None
        L.oracle_plus.argtypes = [dp, dp, dp]
        L.oracle_plus.restype = None
        for n in ("oracle_expmap", "oracle_logmap", "oracle_right_jacobian"):
            getattr(L, n).argtypes = [dp, dp]
            getattr(L, n).restype = None
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(dp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def noise_c(nd):
    nz = capi.ImuNoiseC()
    for k in ("cov_w", "cov_a", "cov_bg", "cov_ba"):
        getattr(nz, k)[:] = list(np.asarray(nd[k], float).ravel())
    return nz


def preintegrate(t, w, a, t_end, bg, ba, noise):
    t, w, a, bg, ba = f64(t), f64(w), f64(a), f64(bg), f64(ba)
    delta, cov, U, jac = np.zeros(11), np.zeros(225), np.zeros(225), np.zeros(45)
    nz = noise_c(noise)
    rc = lib().oracle_preintegrate(len(t), _d(t), _d(w), _d(a), float(t_end), _d(bg), _d(ba), C.byref(nz), _d(delta), _d(cov), _d(U), _d(jac))
    assert rc == 0
    return delta, cov, U, jac


def solve(problem, state, summary):
    pb, st = problem.as_c(), state.as_c()
    rc = lib().oracle_ba_solve(C.byref(pb), C.byref(st), C.byref(summary.c))
    assert rc == 0
    return summary


def set_sum_order(order):
    """0: the reference's order of every sum over landmarks; 1: last landmark first; 2: even landmarks, then odd;
    3: order 2 backwards (tests: what a reordering of the sums is entitled to)"""
    lib().oracle_debug_sum_order(C.c_int32(int(order)))


def linearize(problem, frame_state, rho):
    pb = problem.as_c()
    N, M = problem.n_frames, problem.n_landmarks
    fs, rho = f64(frame_state), f64(rho)
    po, mo = np.zeros(N, np.int32), np.zeros(N, np.int32)
    P = lib().oracle_ba_linearize(C.byref(pb), _d(fs), _d(rho), None, None, None, None, None, None,
                                  po.ctypes.data_as(capi.c_int32_p), mo.ctypes.data_as(capi.c_int32_p))
    cost = np.zeros(1)
    Hpp, gp, Hll, bl, W = np.zeros((P, P)), np.zeros(P), np.zeros(M), np.zeros(M), np.zeros((M, P))
    lib().oracle_ba_linearize(C.byref(pb), _d(fs), _d(rho), _d(cost), _d(Hpp), _d(gp), _d(Hll), _d(bl), _d(W),
                              po.ctypes.data_as(capi.c_int32_p), mo.ctypes.data_as(capi.c_int32_p))
    return dict(cost=cost[0], Hpp=Hpp, gp=gp, Hll=Hll, bl=bl, W=W, pose_off=po, motion_off=mo, P=P)


def cost(problem, frame_state, rho, user=None):
    pb = problem.as_c()
    fs, rho = f64(frame_state), f64(rho)
    u = f64(user) if user is not None else fs
    c = np.zeros(1)
    lib().oracle_ba_cost(C.byref(pb), _d(fs), _d(rho), _d(u), _d(c))
    return c[0]


def marginalize(problem, state, victim, want_info=True):
    pb, st = problem.as_c(), state.as_c()
    n = problem.n_frames - 1
    S, s = np.zeros((15 * n, 15 * n)), np.zeros(15 * n)
    IM, iv = np.zeros((15 * n, 15 * n)), np.zeros(15 * n)
    pr = capi.BAPriorC()
    pr.S, pr.s = _d(S), _d(s)
    if want_info:
        pr.info_matrix, pr.info_vector = _d(IM), _d(iv)
    rc = lib().oracle_ba_marginalize(C.byref(pb), C.byref(st), int(victim), C.byref(pr))
    assert rc == 0
    return S, s, IM, iv


def post_passes(problem, frame_state, t):
    """bundle_adjustor.cpp:251-296 on the flat track table `t` (tests/host_compare.flat_tracks); updates valid / plane / inv_depth /
    quality / membership in place"""
    L = lib()
    pbc = problem.as_c()
    u8p, i32p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    fs = np.ascontiguousarray(frame_state, float)
    L.oracle_post_passes.restype = C.c_int
    rc = L.oracle_post_passes(C.byref(pbc), _d(fs), C.c_int32(len(t["ptr"]) - 1), t["ptr"].ctypes.data_as(i32p), t["frame"].ctypes.data_as(i32p),
                              _d(t["z"]), t["life"].ctypes.data_as(i64p), t["valid"].ctypes.data_as(u8p), t["plane"].ctypes.data_as(u8p),
                              _d(t["inv_depth"]), _d(t["quality"]), C.c_int32(len(t["distance"])), _d(t["normal"]), _d(t["distance"]),
                              t["membership"].ctypes.data_as(u8p))
    assert rc == 0


def reprojection_error(problem, state):
    pb, st = problem.as_c(), state.as_c()
    out = np.zeros(1)
    lib().oracle_ba_reprojection_error(C.byref(pb), C.byref(st), _d(out))
    return out[0]


# ---- KLT front end --------------------------------------------------------------------------------------------
u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


def clahe(img, clip_limit=6.0, tiles=(8, 8)):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros_like(img)
    f = lib().oracle_clahe
    f.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, u8p, C.c_int]
    f.restype = None
    f(img.ctypes.data_as(u8p), w, h, w, float(clip_limit), tiles[0], tiles[1], out.ctypes.data_as(u8p), w)
    return out


def build_pyramid(img):
    """-> list of (u8 image, int16 [h][w][2] Scharr derivative) per level, like buildOpticalFlowPyramid(.., 3, true)."""
    L = lib()
    L.oracle_pyr_down.argtypes = [u8p, C.c_int, C.c_int, u8p]
    L.oracle_pyr_down.restype = None
    L.oracle_scharr.argtypes = [u8p, C.c_int, C.c_int, i16p]
    L.oracle_scharr.restype = None
    L.oracle_pyramid_sizes.argtypes = [C.c_int, C.c_int, i32p, i32p]
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    ws, hs = np.zeros(4, np.int32), np.zeros(4, np.int32)
    n = L.oracle_pyramid_sizes(w, h, ws.ctypes.data_as(i32p), hs.ctypes.data_as(i32p))
    levels = []
    cur = img
    for l in range(n):
        if l > 0:
            nxt = np.zeros((hs[l], ws[l]), np.uint8)
            L.oracle_pyr_down(cur.ctypes.data_as(u8p), int(ws[l - 1]), int(hs[l - 1]), nxt.ctypes.data_as(u8p))
            cur = nxt
        d = np.zeros((hs[l], ws[l], 2), np.int16)
        L.oracle_scharr(cur.ctypes.data_as(u8p), int(ws[l]), int(hs[l]), d.ctypes.data_as(i16p))
        levels.append((cur, d))
    return levels


def klt_track(prev_levels, next_levels, prev_xy, next_xy_init, scalar_order=False):
    """oracle_klt_track: the float sums in the DEFINED order (63 runs of 7 pixels folded by a fixed tree = what klt.hip implements; bit-exact
    contract).  scalar_order=True: OpenCV's scalar left-to-right order (oracle_klt_track_scalar_order), for the order-sensitivity tests."""
    L = lib()
    n_levels = len(prev_levels)
    ws = np.array([lv[0].shape[1] for lv in prev_levels], np.int32)
    hs = np.array([lv[0].shape[0] for lv in prev_levels], np.int32)
    PI = (u8p * n_levels)(*[lv[0].ctypes.data_as(u8p) for lv in prev_levels])
    PD = (i16p * n_levels)(*[lv[1].ctypes.data_as(i16p) for lv in prev_levels])
    NI = (u8p * n_levels)(*[lv[0].ctypes.data_as(u8p) for lv in next_levels])
    prev_xy = np.ascontiguousarray(prev_xy, dtype=np.float32)
    nxt = np.array(next_xy_init, dtype=np.float32, order="C", copy=True)
    n = prev_xy.shape[0]
    status = np.zeros(n, np.uint8)
    fn = L.oracle_klt_track_scalar_order if scalar_order else L.oracle_klt_track
    fn.argtypes = [C.c_int, i32p, i32p, C.POINTER(u8p), C.POINTER(i16p), C.POINTER(u8p), C.c_int, f32p, f32p, u8p]
    fn.restype = None
    fn(n_levels, ws.ctypes.data_as(i32p), hs.ctypes.data_as(i32p), PI, PD, NI, n, prev_xy.ctypes.data_as(f32p),
                       nxt.ctypes.data_as(f32p), status.ctypes.data_as(u8p))
    return nxt, status


def harris_response(img):
    """oracle_harris_response: float32 response map of a (CLAHE'd) u8 image."""
    L = lib()
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    L.oracle_harris_response(img.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(w), C.c_int(h), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def good_features(resp, max_corners=1000, quality=1.0e-3, min_distance=20.0):
    """oracle_good_features: (xy float32 [n, 2], response float32 [n]) in selection order."""
    L = lib()
    L.oracle_good_features.restype = C.c_int
    resp = np.ascontiguousarray(resp, np.float32)
    h, w = resp.shape
    xy = np.zeros((max(max_corners, 1), 2), np.float32)
    r = np.zeros(max(max_corners, 1), np.float32)
    n = L.oracle_good_features(resp.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(w), C.c_int(h), C.c_int(max_corners), C.c_double(quality), C.c_double(min_distance),
                               xy.ctypes.data_as(C.POINTER(C.c_float)), r.ctypes.data_as(C.POINTER(C.c_float)))
    return xy[:n].copy(), r[:n].copy()
