// sym_eig.h -- host-side symmetric eigen-decomposition (Householder tridiagonalization + implicit QL), used once per
// marginalization for the 15(N-1) x 15(N-1) information matrix (replaces Eigen::SelfAdjointEigenSolver at
// pvio/src/pvio/estimation/bundle_adjustor.cpp:584).
//
// Storage is chosen for the QL phase, which is where the time goes (about 2 sweeps x n^2 plane rotations, each over an
// n-vector): eigenvector k is ROW k of Vt, so a rotation mixes two contiguous rows, and every inner loop of the reduction
// and of the accumulation of the reflectors runs over contiguous memory too (the EISPACK column-major organisation with
// the roles of the indices exchanged).  Reductions use four partial sums so that they vectorize without -ffast-math.
// Measured on the GPU box's host, 135 x 135 marginalization matrix: 1000 us with the strided round-2 layout -> see
// profiles/NOTES_r1_r3.md section 5.
#pragma once

namespace pvba {

// A: n x n row-major symmetric (lower triangle is read).  On return w holds the eigenvalues in ascending order and ROW k of
// Vt (n x n row-major) is the unit eigenvector of w[k].  Dispatches once, by CPUID, to the AVX2+FMA build of sym_eig.cpp when the
// host has it, to the baseline x86-64 build otherwise.
void sym_eig(const double *A, int n, double *w, double *Vt);
const char *sym_eig_isa(); // "avx2" | "generic": which build the dispatcher picked (diagnostics)

} // namespace pvba
