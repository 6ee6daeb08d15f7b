"""Does a solve depend on the LDS / register contents it starts on?  For every library given: poison the LDS of all CUs (NaN / 1e300 / 0 / none),
solve, compare with the oracle.  usage: python tests/micro/lds_poison_probe.py lib1.so ..."""
import ctypes as C, os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ba_compare
from oracle import oracle_py as O
from pvio_amd import BAState, BASummary, capi
from pvio_amd.solver import HipContext
O.build()
P = C.CDLL(os.path.join(ROOT, "tests", "micro", "liblds_poison.so"))
P.lds_poison.argtypes = [C.c_uint64]
Rg = C.CDLL(os.path.join(ROOT, "tests", "micro", "libreg_poison.so"))
Rg.reg_poison.argtypes = [C.c_uint64]
REGS = os.environ.get("POISON_REGS", "1") != "0"
pats = {"none": None, "nan": struct.unpack("<Q", struct.pack("<d", float("nan")))[0], "1e300": struct.unpack("<Q", struct.pack("<d", 1e300))[0], "zero": 0,
        "ones": 0x3FF0000000000000}
cases = {"vision_10x200": dict(n_frames=10, n_landmarks=200), "vio_6x40": dict(n_frames=6, n_landmarks=40, use_inertial=True, visibility=4),
         "vio_10x1000": dict(n_frames=10, n_landmarks=1000, use_inertial=True)}
for path in sys.argv[1:]:
    for graph in (False, True):
        ctx = HipContext(lib=capi.load(path), device=0, use_graph=graph)
        for cname, kw in cases.items():
            pb = ba_compare.make(O, **kw)
            st0, sm0 = BAState(pb), BASummary(pb)
            O.solve(pb, st0, sm0)
            row = []
            for pname, pat in pats.items():
                for rep in range(2):
                    if pat is not None:
                        assert P.lds_poison(pat) == 0
                        if REGS:
                            assert Rg.reg_poison(pat) == 0
                    st1, sm1 = ctx.solve(pb)
                    n = min(sm0.trace_len, sm1.trace_len)
                    d = max(float(np.nanmax(np.abs(sm1.trace_states[k] - sm0.trace_states[k]))) if np.isfinite(sm1.trace_states[k]).all() else float("inf") for k in range(n))
                    row.append("%s:%s%.0e/%d" % (pname, "" if sm1.num_iterations == sm0.num_iterations else "ITER!", d, sm1.num_iterations))
            print("%-24s graph=%d %-14s %s" % (os.path.basename(path), graph, cname, "  ".join(row)), flush=True)
        ctx.close()
