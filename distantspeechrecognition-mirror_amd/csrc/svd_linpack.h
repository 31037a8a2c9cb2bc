// csrc/svd_linpack.h -- LINPACK CSVDC restated (complex<float>, job = 11) and the reference's pseudo-inverse built on it
// (btk/beamformer/beamformer.cc:253-305).  Host-only; see svd_linpack.cpp.
#pragma once
#include <complex>

namespace dsr { namespace linpack {
typedef std::complex<float> cf;
// x [ldx][p] column major (destroyed); s, e: at least 2 (n + p) + 2 entries; u [ldu][n], v [ldv][p] column major.  Returns LINPACK's info.
int csvdc(cf* x, int ldx, int n, int p, cf* s, cf* e, cf* u, int ldu, cf* v, int ldv);
// A [M][N] row major -> invA [N][M] row major; false when a singular value fell below dThreshold or the iteration did not converge.
// svals (optional): min(M, N) singular values as csvdc returns them.
bool pseudoinverse(const std::complex<double>* A, std::complex<double>* invA, int M, int N, float dThreshold, float* svals = nullptr);
}}
