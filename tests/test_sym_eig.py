"""The host eigen-solver of the marginalization (pvio_amd/csrc/sym_eig.cpp; replaces Eigen::SelfAdjointEigenSolver at
pvio/src/pvio/estimation/bundle_adjustor.cpp:584): both builds (baseline x86-64, AVX2+FMA) against numpy.linalg.eigh.

Host-only code of the product library: runs without a GPU (the library loads wherever the ROCm runtime is installed)."""
import ctypes as C

import numpy as np
import pytest

from pvio_amd import capi

_DP = C.POINTER(C.c_double)
_BUILDS = {"dispatched": "_ZN4pvba7sym_eigEPKdiPdS2_", "generic": "_ZN4pvba15sym_eig_genericEPKdiPdS2_", "avx2": "_ZN4pvba12sym_eig_avx2EPKdiPdS2_"}


def _has_avx2():
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " avx2" in flags and " fma" in flags


def _eig(build, A):
    lib = C.CDLL(capi.LIB_PATH)
    fn = getattr(lib, _BUILDS[build])
    fn.argtypes, fn.restype = [_DP, C.c_int, _DP, _DP], None
    n = A.shape[0]
    A = np.ascontiguousarray(np.tril(A) + np.triu(np.full_like(A, np.nan), 1))  # only the lower triangle may be read
    w, Vt = np.zeros(n), np.zeros((n, n))
    fn(A.ctypes.data_as(_DP), n, w.ctypes.data_as(_DP), Vt.ctypes.data_as(_DP))
    return w, Vt


def _matrices():
    rng = np.random.default_rng(5)
    out = {}
    for n in (1, 2, 3, 15, 30, 105, 135):
        B = rng.standard_normal((n, n))
        out["information_%d" % n] = B @ np.diag(10.0 ** rng.uniform(-9, 8, n)) @ B.T  # the spread of a marginalization's Schur complement
    Q, _ = np.linalg.qr(rng.standard_normal((45, 45)))
    out["rank_deficient_45"] = Q[:, :30] @ np.diag(rng.uniform(1, 100, 30)) @ Q[:, :30].T
    out["repeated_45"] = Q @ np.diag(np.repeat([1.0, 2.0, 5.0], 15)) @ Q.T
    out["diagonal_20"] = np.diag(rng.uniform(-3, 3, 20))
    out["zero_7"] = np.zeros((7, 7))
    T = np.diag(rng.uniform(1, 2, 40)) + np.diag(rng.uniform(0.1, 1, 39), 1)
    out["tridiagonal_40"] = T + T.T
    B = rng.standard_normal((30, 30))
    out["huge_30"] = (B @ B.T) * 1e150   # squares of the sweep's entries overflow 1e280: the hypot() branch of the rotation radius
    out["tiny_30"] = (B @ B.T) * 1e-150  # ... and underflow 1e-280
    return {k: (v + v.T) / 2 for k, v in out.items()}


@pytest.mark.parametrize("build", sorted(_BUILDS))
@pytest.mark.parametrize("name", sorted(_matrices()))
def test_sym_eig_against_numpy(build, name):
    if build == "avx2" and not _has_avx2():
        pytest.skip("host without AVX2+FMA")
    A = _matrices()[name]
    n = A.shape[0]
    w, Vt = _eig(build, A)
    scale = max(np.abs(A).max(), 1e-300)
    assert np.all(np.diff(w) >= 0), "eigenvalues ascending"
    np.testing.assert_allclose(w, np.linalg.eigh(A)[0], rtol=0, atol=1e-13 * scale * n)
    np.testing.assert_allclose(Vt @ Vt.T, np.eye(n), rtol=0, atol=1e-13 * n)  # rows = unit eigenvectors
    np.testing.assert_allclose(Vt.T @ np.diag(w) @ Vt, A, rtol=0, atol=1e-13 * scale * n)


def test_dispatcher_reports_its_build():
    lib = C.CDLL(capi.LIB_PATH)
    fn = getattr(lib, "_ZN4pvba11sym_eig_isaEv")
    fn.restype = C.c_char_p
    assert fn().decode() == ("avx2" if _has_avx2() else "generic")


@pytest.mark.parametrize("name", sorted(_matrices()))
def test_both_builds_are_bit_identical(name):
    """both builds are compiled -ffp-contract=off, so the instruction set the dispatcher picks changes the speed and nothing else -- the
    prior of a marginalization (and which of its eigenvalues fall under the reference's 1e-8 cut, bundle_adjustor.cpp:586-587) does not depend on the host"""
    if not _has_avx2():
        pytest.skip("host without AVX2+FMA")
    A = _matrices()[name]
    wa, Va = _eig("generic", A)
    wb, Vb = _eig("avx2", A)
    assert (wa == wb).all() and (Va == Vb).all()
