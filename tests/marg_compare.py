"""marginalize_frame parity: C-ABI (emulated build or GPU) vs the oracle.  The sqrt-information factor S is unique only
up to row order / sign (eigenvectors), so the comparison is made on the invariants S^T S, S^T s and on the Schur
complement (information matrix / vector) before the eigen-decomposition."""
import numpy as np

import ba_compare
from pvio_amd import BAState, BASummary


def solved_window(oracle, regular_prior=False, **kw):
    pb = ba_compare.make(oracle, **kw)
    if regular_prior:
        # a well-conditioned prior as produced by earlier marginalizations.  (With the first-time 1e15 gauge prior still
        # in the window -- information 1e30 -- eigenvalues below ~1e14 of the new prior are rounding noise in ANY
        # implementation, the reference's included; that case is only meaningful for victim 0, which removes it.)
        rng = np.random.default_rng(5)
        n = pb.prior_frames.shape[0]
        Q, _ = np.linalg.qr(rng.normal(size=(15 * n, 15 * n)))
        pb.prior_S = np.ascontiguousarray(np.diag(10.0 ** rng.uniform(0.5, 3.0, 15 * n)) @ Q)
        pb.prior_s = rng.normal(size=15 * n)
    st, sm = BAState(pb), BASummary(pb)
    oracle.solve(pb, st, sm)  # marginalization happens at a converged-ish state with non-zero residuals
    return pb, st


def check_marginalize(ctx, oracle, victim, **kw):
    pb, st = solved_window(oracle, regular_prior=(victim != 0), **kw)
    S0, s0, IM0, iv0 = oracle.marginalize(pb, st, victim)
    S1, s1, IM1, iv1 = ctx.marginalize(pb, st, victim)
    scale = np.abs(IM0).max()
    np.testing.assert_allclose(IM1, IM0, rtol=1e-7, atol=1e-9 * scale)
    np.testing.assert_allclose(iv1, iv0, rtol=1e-7, atol=1e-9 * np.abs(iv0).max())
    assert np.abs(IM1 - IM1.T).max() <= 1e-9 * scale
    # S^T S reproduces the information matrix on its numerically non-null part; S^T s the information vector
    np.testing.assert_allclose(S1.T @ S1, S0.T @ S0, rtol=1e-6, atol=1e-7 * scale)
    np.testing.assert_allclose(S1.T @ s1, S0.T @ s0, rtol=1e-6, atol=1e-6 * np.abs(iv0).max())
    # spectrum: the part that is above the rounding noise of the matrix (eigenvalues within ~1e-12 of the largest are
    # noise: whether such a value falls on one or the other side of the reference's absolute 1e-8 cut depends on the
    # summation order and must not be compared) is reproduced by S^T S; everything else stays at noise level
    w = np.linalg.eigvalsh(IM1)
    keep = w > max(1e-8, 1e-11 * scale)
    wS = np.sort(np.linalg.eigvalsh(S1.T @ S1))
    np.testing.assert_allclose(wS[-keep.sum():], w[keep], rtol=1e-6)
    assert np.abs(wS[:len(wS) - keep.sum()]).max(initial=0.0) <= 1e-10 * scale
    # the new prior evaluated at its own linearization point has residual s (marginalization_error_cost.h:91):
    # cost there = |s|^2 / 2, gradient S^T s = projected information vector
    return dict(n=S1.shape[0], rank=int(keep.sum()))
