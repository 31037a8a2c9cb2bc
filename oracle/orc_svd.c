/*
 * oracle/orc_svd.c -- TEST INFRASTRUCTURE (see orc.h).
 * CPU restatement of LINPACK CSVDC as the reference ships it
 *   btk/matrix/linpack_c.cc:9518-10197   csvdc (complex<float>, called with job = 11)
 *   btk/matrix/blas1_c.cc:5,56,108,308,740,855,900,1555  cabs1, cabs2, caxpy, cdotc, cscal, csign2, csrot, scnrm2
 *   btk/matrix/linpack_c.cc:10798        srotg
 * Pinned bit for bit against the reference's own routine compiled from its sources (oracle/_ref,
 * tests/test_oracle_cpu.py::test_csvdc_restatement_matches_reference_bits) and against the committed outputs of that
 * build (tests/golden/linpack_csvdc.npz).
 *
 * Indexing is 1-based through macros, as in the LINPACK Users' Guide.  Arithmetic notes that decide the last bit:
 *   - pow(float, int) in the reference is std::pow -> double: the squares inside scnrm2 and cabs2 are formed in double;
 *   - complex<float> division is libgcc's __divsc3: quotient formed in double by the textbook formula, rounded once;
 *   - the deflation loop of "kase 1" runs m-1 steps from k = m-2+l (the shipped C++ translation), not from k = m-1.
 */
#include "orc.h"
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef float complex cc;

static float r_abs1(cc z) { return fabsf(crealf(z)) + fabsf(cimagf(z)); }
static float r_abs2(cc z) { double a = crealf(z), b = cimagf(z); return (float) sqrt(a * a + b * b); }
static cc r_div(cc x, cc y)
{
  double a = crealf(x), b = cimagf(x), c = crealf(y), d = cimagf(y), den = c * c + d * d;
  return (float) ((a * c + b * d) / den) + I * (float) ((b * c - a * d) / den);
}
static cc r_mul(cc x, cc y)      /* (ac - bd) + i(ad + bc), float products, one rounding each: no library call */
{
  float a = crealf(x), b = cimagf(x), c = crealf(y), d = cimagf(y);
  return (a * c - b * d) + I * (a * d + b * c);
}
static cc r_smul(float s, cc x) { return (s * crealf(x)) + I * (s * cimagf(x)); }
static float r_nrm2(int n, const cc* x)
{
  float scale = 0.0f, ssq = 1.0f;
  if (n < 1) return 0.0f;
  for (int i = 0; i < n; i++) {
    float parts[2] = { crealf(x[i]), cimagf(x[i]) };
    for (int k = 0; k < 2; k++) {
      if (parts[k] == 0.0f) continue;
      float t = fabsf(parts[k]);
      if (scale < t) { double q = scale / t; ssq = (float) (1.0 + ssq * (q * q)); scale = t; }
      else { double q = t / scale; ssq = (float) (ssq + q * q); }
    }
  }
  return scale * sqrtf(ssq);
}
static cc r_sign2(cc z1, cc z2)
{
  float a = r_abs2(z2);
  if (a == 0.0f) return 0.0f;
  cc ph = (crealf(z2) / a) + I * (cimagf(z2) / a);
  return r_smul(r_abs2(z1), ph);
}
static cc r_dotc(int n, const cc* x, const cc* y)
{
  cc v = 0.0f;
  for (int i = 0; i < n; i++) { cc p = r_mul(conjf(x[i]), y[i]); v = (crealf(v) + crealf(p)) + I * (cimagf(v) + cimagf(p)); }
  return v;
}
static void r_axpy(int n, cc a, const cc* x, cc* y)
{
  if (n <= 0 || r_abs1(a) == 0.0f) return;
  for (int i = 0; i < n; i++) { cc p = r_mul(a, x[i]); y[i] = (crealf(y[i]) + crealf(p)) + I * (cimagf(y[i]) + cimagf(p)); }
}
static void r_scal(int n, cc a, cc* x) { for (int i = 0; i < n; i++) x[i] = r_mul(a, x[i]); }
static void r_rot(int n, cc* x, cc* y, float c, float s)
{
  for (int i = 0; i < n; i++) {
    cc cx = r_smul(c, x[i]), sy = r_smul(s, y[i]), cy = r_smul(c, y[i]), sx = r_smul(s, x[i]);
    cc t = (crealf(cx) + crealf(sy)) + I * (cimagf(cx) + cimagf(sy));
    y[i] = (crealf(cy) - crealf(sx)) + I * (cimagf(cy) - cimagf(sx));
    x[i] = t;
  }
}
static void r_rotg(float* sa, float* sb, float* c, float* s)
{
  float roe = (fabsf(*sb) < fabsf(*sa)) ? *sa : *sb;
  float scale = fabsf(*sa) + fabsf(*sb), r, z;
  if (scale == 0.0f) { *c = 1.0f; *s = 0.0f; r = 0.0f; }
  else {
    r = scale * sqrtf((*sa / scale) * (*sa / scale) + (*sb / scale) * (*sb / scale));
    r = (roe < 0.0f ? -1.0f : 1.0f) * r;
    *c = *sa / r; *s = *sb / r;
  }
  if (0.0f < fabsf(*c) && fabsf(*c) <= *s) z = (float) (1.0 / *c); else z = *s;
  *sa = r; *sb = z;
}
static cc c_add(cc a, cc b) { return (crealf(a) + crealf(b)) + I * (cimagf(a) + cimagf(b)); }
static cc c_sub(cc a, cc b) { return (crealf(a) - crealf(b)) + I * (cimagf(a) - cimagf(b)); }

/* x: ldx x p column major (destroyed); s, e: 2 (n + p) + 2 entries; u: ldu x n; v: ldv x p.  interleaved floats. */
int orc_csvdc(float* xf, int ldx, int n, int p, float* sf, float* ef, float* uf, int ldu, float* vf, int ldv)
{
  cc *x = (cc*) xf, *s = (cc*) sf, *e = (cc*) ef, *u = (cc*) uf, *v = (cc*) vf;
#define X(i, j) x[((i) - 1) + (size_t) ((j) - 1) * ldx]
#define U(i, j) u[((i) - 1) + (size_t) ((j) - 1) * ldu]
#define V(i, j) v[((i) - 1) + (size_t) ((j) - 1) * ldv]
#define S(i) s[(i) - 1]
#define E(i) e[(i) - 1]
  const int maxit = 30;
  cc* work = (cc*) calloc((size_t) (n > 0 ? n : 1), sizeof(cc));
  int info = 0;
  const int ncu = n;                                     /* job = 11 */
  int nct = (n - 1 < p) ? n - 1 : p;
  int nrt = (p - 2 < n) ? p - 2 : n; if (nrt < 0) nrt = 0;
  int lu = nct > nrt ? nct : nrt;
  int l, lp1, i, j, k, m, mm, iter, kase, ll, ls = 0, lls, kk;

  for (l = 1; l <= lu; l++) {
    lp1 = l + 1;
    if (l <= nct) {
      S(l) = r_nrm2(n - l + 1, &X(l, l));
      if (r_abs1(S(l)) != 0.0f) {
        if (r_abs1(X(l, l)) != 0.0f) S(l) = r_sign2(S(l), X(l, l));
        r_scal(n - l + 1, r_div(1.0f, S(l)), &X(l, l));
        X(l, l) = c_add(1.0f, X(l, l));
      }
      S(l) = -S(l);
    }
    for (j = lp1; j <= p; j++) {
      if (l <= nct && r_abs1(S(l)) != 0.0f) {
        cc t = r_div(-r_dotc(n - l + 1, &X(l, l), &X(l, j)), X(l, l));
        r_axpy(n - l + 1, t, &X(l, l), &X(l, j));
      }
      E(j) = conjf(X(l, j));
    }
    if (l <= nct) for (i = l; i <= n; i++) U(i, l) = X(i, l);
    if (l <= nrt) {
      E(l) = r_nrm2(p - l, &E(lp1));
      if (r_abs1(E(l)) != 0.0f) {
        if (r_abs1(E(lp1)) != 0.0f) E(l) = r_sign2(E(l), E(lp1));
        r_scal(p - l, r_div(1.0f, E(l)), &E(lp1));
        E(lp1) = c_add(1.0f, E(lp1));
      }
      E(l) = -conjf(E(l));
      if (lp1 <= n && r_abs1(E(l)) != 0.0f) {
        for (j = lp1; j <= n; j++) work[j - 1] = 0.0f;
        for (j = lp1; j <= p; j++) r_axpy(n - l, E(j), &X(lp1, j), &work[lp1 - 1]);
        for (j = lp1; j <= p; j++) r_axpy(n - l, conjf(r_div(-E(j), E(lp1))), &work[lp1 - 1], &X(lp1, j));
      }
      for (i = lp1; i <= p; i++) V(i, l) = E(i);
    }
  }
  m = (p < n + 1) ? p : n + 1;
  if (nct < p) S(nct + 1) = X(nct + 1, nct + 1);
  if (n < m) S(m) = 0.0f;
  if (nrt + 1 < m) E(nrt + 1) = X(nrt + 1, m);
  E(m) = 0.0f;

  for (j = nct + 1; j <= ncu; j++) { for (i = 1; i <= n; i++) U(i, j) = 0.0f; U(j, j) = 1.0f; }
  for (ll = 1; ll <= nct; ll++) {
    l = nct - ll + 1;
    if (r_abs1(S(l)) != 0.0f) {
      for (j = l + 1; j <= ncu; j++) {
        cc t = r_div(-r_dotc(n - l + 1, &U(l, l), &U(l, j)), U(l, l));
        r_axpy(n - l + 1, t, &U(l, l), &U(l, j));
      }
      r_scal(n - l + 1, -1.0f, &U(l, l));
      U(l, l) = c_add(1.0f, U(l, l));
      for (i = 1; i <= l - 1; i++) U(i, l) = 0.0f;
    } else {
      for (i = 1; i <= n; i++) U(i, l) = 0.0f;
      U(l, l) = 1.0f;
    }
  }
  for (ll = 1; ll <= p; ll++) {
    l = p - ll + 1; lp1 = l + 1;
    if (l <= nrt && r_abs1(E(l)) != 0.0f)
      for (j = lp1; j <= p; j++) {
        cc t = r_div(-r_dotc(p - l, &V(lp1, l), &V(lp1, j)), V(lp1, l));
        r_axpy(p - l, t, &V(lp1, l), &V(lp1, j));
      }
    for (i = 1; i <= p; i++) V(i, l) = 0.0f;
    V(l, l) = 1.0f;
  }
  for (i = 1; i <= m; i++) {
    if (r_abs1(S(i)) != 0.0f) {
      cc t = cabsf(S(i)), r = r_div(S(i), t);
      S(i) = t;
      if (i < m) E(i) = r_div(E(i), r);
      r_scal(n, r, &U(1, i));
    }
    if (i == m) break;
    if (r_abs1(E(i)) != 0.0f) {
      cc t = cabsf(E(i)), r = r_div(t, E(i));
      E(i) = t;
      S(i + 1) = r_mul(S(i + 1), r);
      r_scal(p, r, &V(1, i + 1));
    }
  }

  mm = m; iter = 0;
  for (;;) {
    float cs, sn, f, g, t1;
    if (m == 0) break;
    if (maxit <= iter) { info = m; break; }
    for (ll = 1; ll <= m; ll++) {
      l = m - ll;
      if (l == 0) break;
      float test = cabsf(S(l)) + cabsf(S(l + 1));
      float ztest = test + cabsf(E(l));
      if (ztest == test) { E(l) = 0.0f; break; }
    }
    if (l == m - 1) kase = 4;
    else {
      lp1 = l + 1;
      for (lls = lp1; lls <= m + 1; lls++) {
        ls = m - lls + lp1;
        if (ls == l) break;
        float test = 0.0f;
        if (ls != m) test = test + cabsf(E(ls));
        if (ls != l + 1) test = test + cabsf(E(ls - 1));
        float ztest = test + cabsf(S(ls));
        if (ztest == test) { S(ls) = 0.0f; break; }
      }
      if (ls == l) kase = 3; else if (ls == m) kase = 1; else { kase = 2; l = ls; }
    }
    l = l + 1;
    if (kase == 1) {
      int mm1 = m - 1;
      f = crealf(E(m - 1)); E(m - 1) = 0.0f;
      for (kk = 1; kk <= mm1; kk++) {                       /* as shipped: kk from 1, not from l */
        k = mm1 - kk + l;
        t1 = crealf(S(k));
        r_rotg(&t1, &f, &cs, &sn);
        S(k) = t1;
        if (k != l) { f = -sn * crealf(E(k - 1)); E(k - 1) = r_smul(cs, E(k - 1)); }
        if (k <= p) r_rot(p, &V(1, k), &V(1, m), cs, sn);
      }
    } else if (kase == 2) {
      f = crealf(E(l - 1)); E(l - 1) = 0.0f;
      for (k = l; k <= m; k++) {
        t1 = crealf(S(k));
        r_rotg(&t1, &f, &cs, &sn);
        S(k) = t1;
        f = -sn * crealf(E(k));
        E(k) = r_smul(cs, E(k));
        r_rot(n, &U(1, k), &U(1, l - 1), cs, sn);
      }
    } else if (kase == 3) {
      float scale = fmaxf(cabsf(S(m)), fmaxf(cabsf(S(m - 1)), fmaxf(cabsf(E(m - 1)), fmaxf(cabsf(S(l)), cabsf(E(l))))));
      float sm = crealf(S(m)) / scale, smm1 = crealf(S(m - 1)) / scale, emm1 = crealf(E(m - 1)) / scale;
      float sl = crealf(S(l)) / scale, el = crealf(E(l)) / scale;
      float b = (float) (((smm1 + sm) * (smm1 - sm) + emm1 * emm1) / 2.0);
      float c = (sm * emm1) * (sm * emm1);
      float shift = 0.0f;
      if (b != 0.0f || c != 0.0f) {
        shift = sqrtf(b * b + c);
        if (b < 0.0f) shift = -shift;
        shift = c / (b + shift);
      }
      f = (sl + sm) * (sl - sm) + shift;
      g = sl * el;
      for (k = l; k <= m - 1; k++) {
        r_rotg(&f, &g, &cs, &sn);
        if (k != l) E(k - 1) = f;
        f = cs * crealf(S(k)) + sn * crealf(E(k));
        E(k) = c_sub(r_smul(cs, E(k)), r_smul(sn, S(k)));
        g = sn * crealf(S(k + 1));
        S(k + 1) = r_smul(cs, S(k + 1));
        r_rot(p, &V(1, k), &V(1, k + 1), cs, sn);
        r_rotg(&f, &g, &cs, &sn);
        S(k) = f;
        f = cs * crealf(E(k)) + sn * crealf(S(k + 1));
        S(k + 1) = c_add(r_smul(-sn, E(k)), r_smul(cs, S(k + 1)));
        g = sn * crealf(E(k + 1));
        E(k + 1) = r_smul(cs, E(k + 1));
        if (k < n) r_rot(n, &U(1, k), &U(1, k + 1), cs, sn);
      }
      E(m - 1) = f;
      iter = iter + 1;
    } else {
      if (crealf(S(l)) < 0.0f) { S(l) = -S(l); r_scal(p, -1.0f, &V(1, l)); }
      while (l != mm) {
        if (crealf(S(l + 1)) <= crealf(S(l))) break;
        cc t = S(l); S(l) = S(l + 1); S(l + 1) = t;
        if (l < p) for (i = 1; i <= p; i++) { cc q = V(i, l); V(i, l) = V(i, l + 1); V(i, l + 1) = q; }
        if (l < n) for (i = 1; i <= n; i++) { cc q = U(i, l); U(i, l) = U(i, l + 1); U(i, l + 1) = q; }
        l = l + 1;
      }
      iter = 0; m = m - 1;
    }
  }
  free(work);
  return info;
#undef X
#undef U
#undef V
#undef S
#undef E
}

int orc_pseudoinverse_mn(const double* A, int M, int N, double* invA, float thr)
{
  /* beamformer.cc:253-305 for an M x N matrix (A row-major [M][N], invA row-major [N][M]): csvdc(job 11) in complex<float>; singular
     values below the threshold are zeroed and flag failure; invA(j,i) = sum_k v(j,k) s(k) conj(u(i,k)) accumulated in float.  The shipped
     loops run k to N whatever M is; for M < N that reads singular values csvdc never set and columns of u that do not exist -- those terms
     are taken as zero here (only scaling() of a nSource x chanN demixing matrix gets there, beamformer.cc:1862). */
  const int K = M < N ? M : N;
  cc* a = (cc*) malloc(sizeof(cc) * M * N); cc* u = (cc*) calloc((size_t) M * M, sizeof(cc)); cc* v = (cc*) calloc((size_t) N * N, sizeof(cc));
  cc* s = (cc*) calloc((size_t) 2 * (M + N) + 2, sizeof(cc)); cc* e = (cc*) calloc((size_t) 2 * (M + N) + 2, sizeof(cc));
  int ret = 1;
  for (int i = 0; i < M; i++) for (int j = 0; j < N; j++)
    a[i + (size_t) j * M] = (float) A[2 * ((size_t) i * N + j)] + I * (float) A[2 * ((size_t) i * N + j) + 1];
  if (orc_csvdc((float*) a, M, M, N, (float*) s, (float*) e, (float*) u, M, (float*) v, N) != 0) ret = 0;
  for (int k = 0; k < N; k++) {
    if (cabsf(s[k]) < thr) { s[k] = 0.0f; ret = 0; }
    else s[k] = r_div(1.0f, s[k]);
  }
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++) {
      cc x = 0.0f;
      for (int k = 0; k < K; k++) x = c_add(x, r_mul(r_mul(v[j + (size_t) k * N], s[k]), conjf(u[i + (size_t) k * M])));
      invA[2 * ((size_t) j * M + i)] = crealf(x); invA[2 * ((size_t) j * M + i) + 1] = cimagf(x);
    }
  free(a); free(u); free(v); free(s); free(e);
  return ret;
}

int orc_pseudoinverse(const double* A, int n, double* invA, float thr)
{
  /* beamformer.cc:253-305: csvdc(job 11) in complex<float>; singular values below the threshold are zeroed and flag
     failure (the caller then substitutes the identity); invA(j,i) = sum_k v(j,k) s(k) conj(u(i,k)) accumulated in float */
  cc* a = (cc*) malloc(sizeof(cc) * n * n); cc* u = (cc*) calloc((size_t) n * n, sizeof(cc)); cc* v = (cc*) calloc((size_t) n * n, sizeof(cc));
  cc* s = (cc*) calloc((size_t) 4 * n + 2, sizeof(cc)); cc* e = (cc*) calloc((size_t) 4 * n + 2, sizeof(cc));
  int ret = 1;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++)
    a[i + (size_t) j * n] = (float) A[2 * ((size_t) i * n + j)] + I * (float) A[2 * ((size_t) i * n + j) + 1];
  if (orc_csvdc((float*) a, n, n, n, (float*) s, (float*) e, (float*) u, n, (float*) v, n) != 0) ret = 0;
  for (int k = 0; k < n; k++) {
    if (cabsf(s[k]) < thr) { s[k] = 0.0f; ret = 0; }
    else s[k] = r_div(1.0f, s[k]);
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      cc x = 0.0f;
      for (int k = 0; k < n; k++) x = c_add(x, r_mul(r_mul(v[j + (size_t) k * n], s[k]), conjf(u[i + (size_t) k * n])));
      invA[2 * ((size_t) j * n + i)] = crealf(x); invA[2 * ((size_t) j * n + i) + 1] = cimagf(x);
    }
  free(a); free(u); free(v); free(s); free(e);
  return ret;
}
