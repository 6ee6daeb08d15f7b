"""A complete keyframe solve and marginalization THROUGH THE HOST ADAPTER (pvio::BundleAdjustor of pvio_amd/host: flatten the Map,
C ABI, write back), with the adapter's and the library's own PVIO_HIP_TIMING lines on stderr.  GPU box: python tests/prof_keyframe.py"""
import os, sys
os.environ["PVIO_HIP_TIMING"] = "1"
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import host_compare, marg_compare, ba_compare
from pvio_amd import BAState
from oracle import oracle_py as O
O.build()
lib = host_compare.load("libpvio_host.so")
for nf, nl in ((10, 1000), (8, 300)):
    pb, st = marg_compare.solved_window(O, regular_prior=False, n_frames=nf, n_landmarks=nl, use_inertial=True)
    sys.stderr.write("---- %d x %d ----\n" % (nf, nl))
    for rep in range(6):
        st1 = BAState(pb)
        host_compare.roundtrip_solve(lib, pb, st1)
    import ctypes as C
    from pvio_amd import capi
    n = nf - 1
    S1, s1 = np.zeros((15 * n, 15 * n)), np.zeros(15 * n)
    for rep in range(6):
        pbc, stc = pb.as_c(), st.as_c()
        lib.host_roundtrip_marginalize(C.byref(pbc), C.byref(stc), C.c_int32(0), S1.ctypes.data_as(capi.c_double_p), s1.ctypes.data_as(capi.c_double_p))
