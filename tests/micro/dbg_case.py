"""Oracle and GPU traces of the named windows (tests/ba_compare.CASES) side by side.  PVIO_LIB=path picks a build of the library,
PVIO_HIP_DEBUG_CTRL=1 makes the solver print the control block after every kernel of the first (plain-launch) solve.
usage: python tests/micro/dbg_case.py CASE [CASE ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import ba_compare
from oracle import oracle_py as O
from pvio_amd import BAState, BASummary
from pvio_amd.solver import HipContext
O.build()
from pvio_amd import capi
libp = os.environ.get('PVIO_LIB')
ctx = HipContext(lib=capi.load(libp), device=0) if libp else HipContext(device=0)
for name in sys.argv[1:]:
    pb = ba_compare.make(O, **ba_compare.CASES[name])
    st0, sm0 = BAState(pb), BASummary(pb); O.solve(pb, st0, sm0)
    st1, sm1 = ctx.solve(pb)
    print(name, "oracle term/it", sm0.termination, sm0.num_iterations, "gpu", sm1.termination, sm1.num_iterations)
    for a, b in list(zip(sm0.trace(), sm1.trace()))[:2]:
        print("  it %d cost %.9e | %.9e  ok %d|%d succ %d|%d radius %.3e|%.3e gmax %.3e|%.3e" % (a["iteration"], a["cost"], b["cost"], a["step_is_valid"], b["step_is_valid"], a["step_is_successful"], b["step_is_successful"], a["trust_region_radius"], b["trust_region_radius"], a["gradient_max_norm"], b["gradient_max_norm"]))
    print("  trace lens", len(sm0.trace()), len(sm1.trace()))
