// oracle/ref_linpack_wrap.cc -- TEST INFRASTRUCTURE.
// C-ABI doorway onto the reference's in-tree LINPACK csvdc
// (/root/reference/btk/matrix/linpack_c.cc:9518), compiled from the reference sources
// where they lie (see Makefile target _ref).  Nothing from the reference is copied:
// this file only forwards the call.  Used by tests to pin the oracle's complex<float>
// pseudo-inverse (beamformer.cc:253-305 builds pinv from exactly this routine).
#include <complex>
#include "linpack_c.H"

extern "C" int ref_csvdc(float* a /*interleaved, column major*/, int lda, int m, int n,
                         float* s, float* e, float* u, int ldu, float* v, int ldv, int job)
{
  return csvdc(reinterpret_cast<std::complex<float>*>(a), lda, m, n,
               reinterpret_cast<std::complex<float>*>(s), reinterpret_cast<std::complex<float>*>(e),
               reinterpret_cast<std::complex<float>*>(u), ldu,
               reinterpret_cast<std::complex<float>*>(v), ldv, job);
}
