"""Host adapter test: flat window -> pvio::Map object graph -> pvio::BundleAdjustor (pvio_amd/host) -> C-ABI -> back.
The adapter re-flattens the Map in the reference's block order, so the result must equal the oracle's up to summation
order."""
import ctypes as C
import os
import subprocess

import numpy as np

import ba_compare
from pvio_amd import BAState, BASummary, capi
from oracle import oracle_py

HOST_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")


def load(target):
    subprocess.check_call(["make", "-s", "-C", HOST_DIR, target])
    lib = C.CDLL(os.path.join(HOST_DIR, target))
    lib.host_roundtrip_solve.restype = C.c_int
    lib.host_roundtrip_marginalize.restype = C.c_int
    return lib


def roundtrip_solve(lib, pb, st, tracks=None):
    """tracks: optional dict that receives the per-track flags / plane memberships after the solve (post-solve passes)"""
    pbc, stc = pb.as_c(), st.as_c()
    N = pb.n_frames
    ptr = np.zeros(N + 1, np.int32)
    ts, ws, as_, tend = [], [], [], np.zeros(N)
    nzc = None
    if pb.use_inertial:
        for j, (t, w, a, te) in enumerate(pb.meta["imu"], start=1):
            ptr[j] = sum(len(x) for x in ts)
            ts.append(t), ws.append(w), as_.append(a)
            tend[j] = te
        ptr[N] = sum(len(x) for x in ts)
        nzc = oracle_py.noise_c(pb.meta["imu_noise"])
    T = np.ascontiguousarray(np.concatenate(ts)) if ts else np.zeros(1)
    W = np.ascontiguousarray(np.concatenate(ws)) if ws else np.zeros(3)
    A = np.ascontiguousarray(np.concatenate(as_)) if as_ else np.zeros(3)
    dp = capi.c_double_p
    usable = C.c_int32(0)
    args = [C.byref(pbc), C.byref(stc), ptr.ctypes.data_as(capi.c_int32_p), T.ctypes.data_as(dp), W.ctypes.data_as(dp),
            A.ctypes.data_as(dp), tend.ctypes.data_as(dp), C.byref(nzc) if nzc is not None else None, C.c_double(1.0e-4), C.byref(usable)]
    if tracks is None:
        rc = lib.host_roundtrip_solve(*args)
    else:
        nt = pb.n_landmarks + pb.n_plane_factors
        u8p = C.POINTER(C.c_uint8)
        tracks.update(valid=np.zeros(nt, np.uint8), plane=np.zeros(nt, np.uint8), inv_depth=np.zeros(nt), quality=np.zeros(nt),
                      membership=np.zeros((8, nt), np.uint8))
        npl = C.c_int32(0)
        lib.host_roundtrip_solve_tracks.restype = C.c_int
        rc = lib.host_roundtrip_solve_tracks(*args, tracks["valid"].ctypes.data_as(u8p), tracks["plane"].ctypes.data_as(u8p),
                                             tracks["inv_depth"].ctypes.data_as(dp), tracks["quality"].ctypes.data_as(dp),
                                             tracks["membership"].ctypes.data_as(u8p), C.byref(npl))
        tracks["membership"] = tracks["membership"].reshape(-1)[:npl.value * nt].reshape(npl.value, nt)
    assert rc == 0
    return usable.value


def flat_tracks(pb, st):
    """Every track of the Map the harness builds from `pb` (landmarks, then plane tracks), as the oracle's post-pass input:
    observation lists with the anchor first, flags before the passes, plane table and memberships."""
    M, P = pb.n_landmarks, pb.n_plane_factors
    ptr, frames, zs = [0], [], []
    for l in range(M):
        frames.append(int(pb.lm_anchor_frame[l])), zs.append(pb.lm_anchor_z[l])
        for o in range(pb.lm_obs_ptr[l], pb.lm_obs_ptr[l + 1]):
            frames.append(int(pb.obs_frame[o])), zs.append(pb.obs_z[o])
        ptr.append(len(frames))
    keys, plane_of = [], []
    for f in range(P):
        for o in range(pb.plane_obs_ptr[f], pb.plane_obs_ptr[f + 1]):
            frames.append(int(pb.plane_obs_frame[o])), zs.append(pb.plane_obs_z[o])
        ptr.append(len(frames))
        key = (tuple(pb.plane_normal[f]), float(pb.plane_distance[f]))
        if key not in keys:
            keys.append(key)
        plane_of.append(keys.index(key))
    nt = M + P
    t = dict(ptr=np.array(ptr, np.int32), frame=np.array(frames, np.int32), z=np.ascontiguousarray(np.array(zs, float).reshape(-1, 2)),
             life=np.diff(ptr).astype(np.int64), valid=np.r_[np.ones(M, np.uint8), np.zeros(P, np.uint8)],
             plane=np.r_[np.zeros(M, np.uint8), np.ones(P, np.uint8)], inv_depth=np.r_[st.lm_inv_depth, np.ones(P)], quality=np.zeros(nt),
             normal=np.ascontiguousarray(np.array([k[0] for k in keys], float).reshape(-1, 3)), distance=np.array([k[1] for k in keys], float),
             membership=np.zeros((len(keys), nt), np.uint8))
    for f in range(P):
        t["membership"][plane_of[f], M + f] = 1
    return t


def check_adapter_post_passes(lib, oracle, **kw):
    """bundle_adjustor.cpp:251-296 through the adapter against the oracle's restatement (oracle/oracle_post.cpp), on a window whose
    planes hold a few tracks that are really 0.3 m off (synth plane_outliers)."""
    pb = ba_compare.make(oracle, **kw)
    st0, sm0 = BAState(pb), BASummary(pb)
    oracle.solve(pb, st0, sm0)
    exp = flat_tracks(pb, st0)
    oracle.post_passes(pb, st0.frame_state, exp)
    st1, got = BAState(pb), {}
    roundtrip_solve(lib, pb, st1, got)
    M = pb.n_landmarks
    assert (got["valid"] == exp["valid"]).all() and (got["plane"] == exp["plane"]).all()
    assert (got["membership"] == exp["membership"]).all()
    moved = (exp["plane"][M:] == 0) & (exp["valid"][M:] == 1)
    np.testing.assert_allclose(got["inv_depth"], exp["inv_depth"], rtol=1e-6, atol=1e-9)
    ok = exp["valid"] == 1
    np.testing.assert_allclose(got["quality"][ok], exp["quality"][ok], rtol=0, atol=1e-5)
    return moved, exp["membership"].sum(0)[M:] == 0


def check_adapter(lib, oracle, **kw):
    pb = ba_compare.make(oracle, **kw)
    st0, sm0 = BAState(pb), BASummary(pb)
    oracle.solve(pb, st0, sm0)
    st1 = BAState(pb)
    usable = roundtrip_solve(lib, pb, st1)
    assert usable == sm0.is_usable
    np.testing.assert_allclose(st1.frame_state, st0.frame_state, rtol=0, atol=1e-6)
    np.testing.assert_allclose(st1.lm_inv_depth, st0.lm_inv_depth, rtol=0, atol=1e-6)
    np.testing.assert_allclose(st1.lm_quality, st0.lm_quality, rtol=0, atol=1e-5)
    assert (st1.lm_valid == st0.lm_valid).all()
    return float(np.abs(st1.frame_state - st0.frame_state).max())


def check_adapter_marginalize(lib, oracle, victim, **kw):
    import marg_compare
    pb, st = marg_compare.solved_window(oracle, regular_prior=(victim != 0), **kw)
    S0, s0, IM0, iv0 = oracle.marginalize(pb, st, victim)
    n = pb.n_frames - 1
    S1, s1 = np.zeros((15 * n, 15 * n)), np.zeros(15 * n)
    pbc, stc = pb.as_c(), st.as_c()
    rc = lib.host_roundtrip_marginalize(C.byref(pbc), C.byref(stc), C.c_int32(victim), S1.ctypes.data_as(capi.c_double_p), s1.ctypes.data_as(capi.c_double_p))
    assert rc == 0
    scale = np.abs(IM0).max()
    np.testing.assert_allclose(S1.T @ S1, S0.T @ S0, rtol=1e-6, atol=1e-7 * scale)
    np.testing.assert_allclose(S1.T @ s1, S0.T @ s0, rtol=1e-6, atol=1e-6 * np.abs(iv0).max())
