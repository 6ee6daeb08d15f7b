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


def roundtrip_solve(lib, pb, st):
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
    rc = lib.host_roundtrip_solve(C.byref(pbc), C.byref(stc), ptr.ctypes.data_as(capi.c_int32_p), T.ctypes.data_as(dp), W.ctypes.data_as(dp),
                                  A.ctypes.data_as(dp), tend.ctypes.data_as(dp), C.byref(nzc) if nzc is not None else None,
                                  C.c_double(1.0e-4), C.byref(usable))
    assert rc == 0
    return usable.value


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
