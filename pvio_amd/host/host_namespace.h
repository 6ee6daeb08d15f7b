// host_namespace.h -- every header under pvio_amd/host/ that declares into `namespace pvio` WITHOUT including a reference header starts here.
// Inside the PVIO tree the namespace is macro-renamed (`#define pvio pvio_0_3_0`, pvio/cmake/version.h.in:27): a translation unit that does not
// see that macro defines its symbols in plain `pvio::` and the ones that do cannot link against them (solve_pnp: pnp.cpp vs pnp_solve.cpp --
// found by LINKING and running the drop-in on the reference's real Map, oracle/ref/Makefile `dropin`; a compile-only check cannot see it).
#pragma once
#ifdef PVIO_HOST_USE_REFERENCE_TYPES
#include <pvio/version.h>
#endif
