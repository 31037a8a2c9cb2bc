// csrc/svd_linpack.cpp -- single-precision complex singular value decomposition after LINPACK's CSVDC
// (Dongarra, Moler, Bunch, Stewart: LINPACK Users' Guide, SIAM 1979, chapter 11) and the pseudo-inverse the
// reference builds from it (btk/beamformer/beamformer.cc:253-305 calls csvdc of btk/matrix/linpack_c.cc:9518
// with job = 11 and assembles V diag(1/s) U^H).
//
// Why a LINPACK restatement and not a Jacobi SVD: the MVDR weights (beamformer.cc:2392-2446) and the
// Lefkimmiatis post-filter's d^H pinv(R) d (postfilter.cc:989-991) inherit the rounding of this fp32 routine on
// badly conditioned coherence matrices (diffuse field at low frequencies: condition 1e5..1e7).  Two different
// fp32 SVD algorithms agree there to 1e-3 only; the same algorithm in the same operation order agrees to the
// last bits.  Host-side set-up code (once per array geometry), compiled with -ffp-contract=off.
//
// Algorithm: Householder reduction to bidiagonal form (columns, then rows), accumulation of U and V, phases
// rotated out so that the bidiagonal is real, then implicit-shift QR sweeps on the real bidiagonal with the four
// LINPACK cases (deflate negligible s[m], split at negligible s[l], QR step, convergence + ordering).
#include "svd_linpack.h"
#include <cmath>
#include <vector>

namespace dsr { namespace linpack {

// |re| + |im| (CABS1) and the Euclidean modulus formed in double and rounded once (CABS2 of the C++ LINPACK)
static inline float abs1(cf z) { return std::fabs(z.real()) + std::fabs(z.imag()); }
static inline float abs2(cf z)
{ const double re = (double) z.real(), im = (double) z.imag(); return (float) std::sqrt(re * re + im * im); }

// complex<float> / complex<float> as the reference build performs it: libgcc's __divsc3 forms the quotient in double with the textbook
// formula and rounds once.  Spelled out so that the result does not depend on which runtime the final link binds (compiler-rt's __divsc3
// scales by logb and rounds differently in the last place).
static inline cf cdiv(cf x, cf y)
{
  const double a = x.real(), b = x.imag(), c = y.real(), d = y.imag(), den = c * c + d * d;
  return cf((float) ((a * c + b * d) / den), (float) ((b * c - a * d) / den));
}
// SCNRM2: scaled sum of squares over real and imaginary parts in turn; the ratio is squared in double
static float nrm2(int n, const cf* x)
{
  if (n < 1) return 0.0f;
  float scale = 0.0f, ssq = 1.0f;
  for (int i = 0; i < n; i++)
    for (int part = 0; part < 2; part++) {
      const float v = part ? x[i].imag() : x[i].real();
      if (v == 0.0f) continue;
      const float a = std::fabs(v);
      if (scale < a) { const double q = (double) (scale / a); ssq = (float) (1.0 + (double) ssq * (q * q)); scale = a; }
      else { const double q = (double) (a / scale); ssq = (float) ((double) ssq + q * q); }
    }
  return scale * std::sqrt(ssq);
}
// CSIGN2: |z1| with the phase of z2
static inline cf sign2(cf z1, cf z2) { const float a = abs2(z2); return a == 0.0f ? cf(0.0f, 0.0f) : abs2(z1) * (z2 / a); }
static inline cf dotc(int n, const cf* x, const cf* y) { cf v(0.0f, 0.0f); for (int i = 0; i < n; i++) v = v + std::conj(x[i]) * y[i]; return v; }
static inline void axpy(int n, cf a, const cf* x, cf* y) { if (n <= 0 || abs1(a) == 0.0f) return; for (int i = 0; i < n; i++) y[i] = y[i] + a * x[i]; }
static inline void scal(int n, cf a, cf* x) { for (int i = 0; i < n; i++) x[i] = a * x[i]; }
static inline void rot(int n, cf* x, cf* y, float c, float s)
{ for (int i = 0; i < n; i++) { const cf t = c * x[i] + s * y[i]; y[i] = c * y[i] - s * x[i]; x[i] = t; } }
// SROTG: Givens rotation that annihilates b; a <- r
static void rotg(float& a, float& b, float& c, float& s)
{
  const float roe = std::fabs(b) < std::fabs(a) ? a : b;
  const float scale = std::fabs(a) + std::fabs(b);
  float r;
  if (scale == 0.0f) { c = 1.0f; s = 0.0f; r = 0.0f; }
  else {
    r = scale * std::sqrt((a / scale) * (a / scale) + (b / scale) * (b / scale));
    r = (roe < 0.0f ? -1.0f : 1.0f) * r;
    c = a / r; s = b / r;
  }
  const float z = (0.0f < std::fabs(c) && std::fabs(c) <= s) ? (float) (1.0 / c) : s;
  a = r; b = z;
}

int csvdc(cf* x, int ldx, int n, int p, cf* s, cf* e, cf* u, int ldu, cf* v, int ldv)
{
  const int maxit = 30;
  std::vector<cf> work(n > 0 ? n : 1);
  auto X = [&](int i, int j) -> cf& { return x[i + (size_t) j * ldx]; };
  auto U = [&](int i, int j) -> cf& { return u[i + (size_t) j * ldu]; };
  auto V = [&](int i, int j) -> cf& { return v[i + (size_t) j * ldv]; };
  const cf one(1.0f, 0.0f), zero(0.0f, 0.0f);
  const int ncu = n;                                         // job = 11: all n left vectors, and the right vectors
  const int nct = std::min(n - 1, p), nrt = std::max(0, std::min(p - 2, n)), lu = std::max(nct, nrt);

  // ---- bidiagonalisation: column l gets a Householder vector (stored in x, copied to u), row l one (in e, copied to v)
  for (int l = 0; l < lu; l++) {
    if (l < nct) {
      s[l] = cf(nrm2(n - l, &X(l, l)), 0.0f);
      if (abs1(s[l]) != 0.0f) {
        if (abs1(X(l, l)) != 0.0f) s[l] = sign2(s[l], X(l, l));
        scal(n - l, cdiv(one, s[l]), &X(l, l));
        X(l, l) = one + X(l, l);
      }
      s[l] = -s[l];
    }
    for (int j = l + 1; j < p; j++) {
      if (l < nct && abs1(s[l]) != 0.0f) {
        const cf t = cdiv(-dotc(n - l, &X(l, l), &X(l, j)), X(l, l));
        axpy(n - l, t, &X(l, l), &X(l, j));
      }
      e[j] = std::conj(X(l, j));                             // row l for the row transformation below
    }
    if (l < nct) for (int i = l; i < n; i++) U(i, l) = X(i, l);
    if (l < nrt) {
      e[l] = cf(nrm2(p - l - 1, &e[l + 1]), 0.0f);
      if (abs1(e[l]) != 0.0f) {
        if (abs1(e[l + 1]) != 0.0f) e[l] = sign2(e[l], e[l + 1]);
        scal(p - l - 1, cdiv(one, e[l]), &e[l + 1]);
        e[l + 1] = one + e[l + 1];
      }
      e[l] = -std::conj(e[l]);
      if (l + 1 < n && abs1(e[l]) != 0.0f) {
        for (int j = l + 1; j < n; j++) work[j] = zero;
        for (int j = l + 1; j < p; j++) axpy(n - l - 1, e[j], &X(l + 1, j), &work[l + 1]);
        for (int j = l + 1; j < p; j++) axpy(n - l - 1, std::conj(cdiv(-e[j], e[l + 1])), &work[l + 1], &X(l + 1, j));
      }
      for (int i = l + 1; i < p; i++) V(i, l) = e[i];
    }
  }

  // ---- the final bidiagonal matrix of order m
  int m = std::min(p, n + 1);
  if (nct < p) s[nct] = X(nct, nct);
  if (n < m) s[m - 1] = zero;
  if (nrt + 1 < m) e[nrt] = X(nrt, m - 1);
  e[m - 1] = zero;

  // ---- U from the stored column reflectors (last to first)
  for (int j = nct; j < ncu; j++) { for (int i = 0; i < n; i++) U(i, j) = zero; U(j, j) = one; }
  for (int l = nct - 1; l >= 0; l--) {
    if (abs1(s[l]) != 0.0f) {
      for (int j = l + 1; j < ncu; j++) {
        const cf t = cdiv(-dotc(n - l, &U(l, l), &U(l, j)), U(l, l));
        axpy(n - l, t, &U(l, l), &U(l, j));
      }
      scal(n - l, cf(-1.0f, 0.0f), &U(l, l));
      U(l, l) = one + U(l, l);
      for (int i = 0; i < l; i++) U(i, l) = zero;
    } else {
      for (int i = 0; i < n; i++) U(i, l) = zero;
      U(l, l) = one;
    }
  }
  // ---- V from the stored row reflectors
  for (int l = p - 1; l >= 0; l--) {
    if (l < nrt && abs1(e[l]) != 0.0f)
      for (int j = l + 1; j < p; j++) {
        const cf t = cdiv(-dotc(p - l - 1, &V(l + 1, l), &V(l + 1, j)), V(l + 1, l));
        axpy(p - l - 1, t, &V(l + 1, l), &V(l + 1, j));
      }
    for (int i = 0; i < p; i++) V(i, l) = zero;
    V(l, l) = one;
  }

  // ---- rotate the phases out of s and e
  for (int i = 0; i < m; i++) {
    if (abs1(s[i]) != 0.0f) {
      const cf t(std::abs(s[i]), 0.0f), r = cdiv(s[i], t);
      s[i] = t;
      if (i + 1 < m) e[i] = cdiv(e[i], r);
      scal(n, r, &U(0, i));
    }
    if (i + 1 == m) break;
    if (abs1(e[i]) != 0.0f) {
      const cf t(std::abs(e[i]), 0.0f), r = cdiv(t, e[i]);
      e[i] = t;
      s[i + 1] = s[i + 1] * r;
      scal(p, r, &V(0, i + 1));
    }
  }

  // ---- QR iteration on the real bidiagonal (1-based l, m as in the Users' Guide; entries are s[l-1], e[l-1])
  const int mm = m; int iter = 0, info = 0;
  while (m != 0) {
    if (iter >= maxit) { info = m; break; }
    int l, kase;
    for (l = m - 1; l >= 1; l--) {
      const float test = std::abs(s[l - 1]) + std::abs(s[l]);
      const float ztest = test + std::abs(e[l - 1]);
      if (ztest == test) { e[l - 1] = zero; break; }
    }
    if (l == m - 1) kase = 4;
    else {
      int ls;
      for (ls = m; ls > l; ls--) {
        float test = 0.0f;
        if (ls != m) test = test + std::abs(e[ls - 1]);
        if (ls != l + 1) test = test + std::abs(e[ls - 2]);
        const float ztest = test + std::abs(s[ls - 1]);
        if (ztest == test) { s[ls - 1] = zero; break; }
      }
      if (ls == l) kase = 3; else if (ls == m) kase = 1; else { kase = 2; l = ls; }
    }
    l = l + 1;
    float cs, sn;
    if (kase == 1) {                                         // deflate negligible s[m]
      float f = e[m - 2].real(); e[m - 2] = zero;
      // The Users' Guide runs k = m-1 down to l.  The C++ LINPACK the reference ships (linpack_c.cc, "for kk = 1 .. mm1; k = mm1 - kk + l")
      // runs m-1 steps from k = m-2+l instead: identical for l = 1, and for l > 1 it also rotates entries at and above m.  Restated as
      // shipped (s and e are sized by the caller for it, as the reference's are); columns of V that do not exist are not touched.
      for (int kk = 1; kk <= m - 1; kk++) {
        const int k = (m - 1) - kk + l;
        float t1 = s[k - 1].real();
        rotg(t1, f, cs, sn);
        s[k - 1] = cf(t1, 0.0f);
        if (k != l) { f = -sn * e[k - 2].real(); e[k - 2] = cs * e[k - 2]; }
        if (k - 1 < p) rot(p, &V(0, k - 1), &V(0, m - 1), cs, sn);
      }
    } else if (kase == 2) {                                  // split at negligible s[l]
      float f = e[l - 2].real(); e[l - 2] = zero;
      for (int k = l; k <= m; k++) {
        float t1 = s[k - 1].real();
        rotg(t1, f, cs, sn);
        s[k - 1] = cf(t1, 0.0f);
        f = -sn * e[k - 1].real();
        e[k - 1] = cs * e[k - 1];
        rot(n, &U(0, k - 1), &U(0, l - 2), cs, sn);
      }
    } else if (kase == 3) {                                  // one implicit-shift QR step
      const float scale = std::max(std::abs(s[m - 1]), std::max(std::abs(s[m - 2]), std::max(std::abs(e[m - 2]),
                          std::max(std::abs(s[l - 1]), std::abs(e[l - 1])))));
      const float sm = s[m - 1].real() / scale, smm1 = s[m - 2].real() / scale, emm1 = e[m - 2].real() / scale;
      const float sl = s[l - 1].real() / scale, el = e[l - 1].real() / scale;
      const float b = (float) ((double) ((smm1 + sm) * (smm1 - sm) + emm1 * emm1) / 2.0);
      const float c = (sm * emm1) * (sm * emm1);
      float shift = 0.0f;
      if (b != 0.0f || c != 0.0f) {
        shift = std::sqrt(b * b + c);
        if (b < 0.0f) shift = -shift;
        shift = c / (b + shift);
      }
      float f = (sl + sm) * (sl - sm) + shift, g = sl * el;
      for (int k = l; k <= m - 1; k++) {                     // chase the bulge
        rotg(f, g, cs, sn);
        if (k != l) e[k - 2] = cf(f, 0.0f);
        f = cs * s[k - 1].real() + sn * e[k - 1].real();
        e[k - 1] = cs * e[k - 1] - sn * s[k - 1];
        g = sn * s[k].real();
        s[k] = cs * s[k];
        rot(p, &V(0, k - 1), &V(0, k), cs, sn);
        rotg(f, g, cs, sn);
        s[k - 1] = cf(f, 0.0f);
        f = cs * e[k - 1].real() + sn * s[k].real();
        s[k] = -sn * e[k - 1] + cs * s[k];
        g = sn * e[k].real();
        e[k] = cs * e[k];
        if (k < n) rot(n, &U(0, k - 1), &U(0, k), cs, sn);
      }
      e[m - 2] = cf(f, 0.0f);
      iter++;
    } else {                                                 // convergence: sign, then order
      if (s[l - 1].real() < 0.0f) { s[l - 1] = -s[l - 1]; scal(p, cf(-1.0f, 0.0f), &V(0, l - 1)); }
      while (l != mm) {
        if (s[l].real() <= s[l - 1].real()) break;
        std::swap(s[l - 1], s[l]);
        if (l < p) for (int i = 0; i < p; i++) std::swap(V(i, l - 1), V(i, l));
        if (l < n) for (int i = 0; i < n; i++) std::swap(U(i, l - 1), U(i, l));
        l++;
      }
      iter = 0; m--;
    }
  }
  return info;
}

// beamformer.cc:253-305 -- A (M x N, row major complex double) -> invA (N x M, row major): V diag(1/s) U^H in complex<float>,
// singular values below dThreshold zeroed (return false, as when csvdc reports non-convergence)
bool pseudoinverse(const std::complex<double>* A, std::complex<double>* invA, int M, int N, float dThreshold, float* svals)
{
  std::vector<cf> a((size_t) M * N), s(2 * ((size_t) M + N) + 2), e(2 * ((size_t) M + N) + 2), u((size_t) M * M), v((size_t) N * N);
  for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) a[i + (size_t) j * M] = cf((float) A[(size_t) i * N + j].real(), (float) A[(size_t) i * N + j].imag());
  bool ret = csvdc(a.data(), M, M, N, s.data(), e.data(), u.data(), M, v.data(), N) == 0;
  if (svals) for (int k = 0; k < std::min(M, N); k++) svals[k] = s[k].real();
  for (int k = 0; k < N; k++) {
    if (std::abs(s[k]) < dThreshold) { s[k] = cf(0.0f, 0.0f); ret = false; }
    else s[k] = cdiv(cf(1.0f, 0.0f), s[k]);
  }
  // (the shipped loop runs k to N whatever M is; for M < N that reads singular values csvdc never set and columns of u that do not exist --
  // only scaling() of an nSource x chanN demixing matrix gets there, beamformer.cc:1862 -- those terms are zero here)
  const int K = std::min(M, N);
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++) {
      cf xacc(0.0f, 0.0f);
      for (int k = 0; k < K; k++) xacc = xacc + v[j + (size_t) k * N] * s[k] * std::conj(u[i + (size_t) k * M]);
      invA[(size_t) j * M + i] = std::complex<double>(xacc.real(), xacc.imag());
    }
  return ret;
}

}}  // namespace dsr::linpack
