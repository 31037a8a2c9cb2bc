/*
 * oracle/orc_mmi.c -- TEST INFRASTRUCTURE (see orc.h).
 * CPU restatement of SubbandMMI (btk/beamformer/beamformer.h:264-312, beamformer.cc:1753-2319) and of what it calls:
 *   putInverseMat22            beamformer.cc:202-242      calcNullBeamformer        :314-394
 *   _calcBlockingMatrix (NC>=1) :398-479                   calcMainlobe / N          :531-594, 631-753
 *   calcSidelobeCancellerP_f/U_f :761-799                  scaling                   :1455-1490
 *   calcOutputOfGSC            :1251-1287                  ZelinskiFilter(_f)        btk/postfilter/postfilter.cc:59-221
 * Quirks of the shipped code are kept and marked "as shipped".  Complex arithmetic follows GSL's formulas (gsl_complex_mul / _div /
 * _abs); the BLAS level-1/2/3 calls are plain sums in index order (which BLAS the reference links against is not fixed, so parity with
 * the product is asserted to 1e-9 relative, not bitwise).  The TYPE_APAB post-filter branch (:2047-2049, 2177-2179) is ApabFilter /
 * ApabFilter_f of postfilter.cc:225-340 with channelX = chanN/2: it filters bins < fftLen/2 only, so its output vector is not
 * conjugate-symmetric without halfBandShift (the product hands over all fftLen bins in that configuration).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <complex.h>

typedef double complex zc;

static inline zc gmul(zc a, zc b) { return (creal(a) * creal(b) - cimag(a) * cimag(b)) + I * (creal(a) * cimag(b) + cimag(a) * creal(b)); }
static inline zc gdiv(zc a, zc b)
{
  const double s = 1.0 / hypot(creal(b), cimag(b)); const double sbr = s * creal(b), sbi = s * cimag(b);
  return ((creal(a) * sbr + cimag(a) * sbi) * s) + I * ((cimag(a) * sbr - creal(a) * sbi) * s);
}
static inline zc gpolar(double r, double th) { return (r * cos(th)) + I * (r * sin(th)); }
static inline zc gmulr(zc a, double x) { return (creal(a) * x) + I * (cimag(a) * x); }
static inline zc gdivr(zc a, double x) { return (creal(a) / x) + I * (cimag(a) / x); }
static inline double gabs2(zc a) { return creal(a) * creal(a) + cimag(a) * cimag(a); }

typedef struct { zc *wq, *B, *wa, *wl, *ta, *csd; } src_t;
struct orc_mmi {
  int M, C, hbs, target, nSource, pfType, NC, haveW, frameX;
  double alpha;
  int useMask; double avgFactor; unsigned fwidth, maskType;
  zc** itf; zc* avgOut; zc* wp1; zc* vec;
  src_t* src;
};

static void free_weights(orc_mmi* m)
{
  if (!m->src) return;
  for (int s = 0; s < m->nSource; s++) { free(m->src[s].wq); free(m->src[s].B); free(m->src[s].wa); free(m->src[s].wl); free(m->src[s].ta); free(m->src[s].csd); }
  free(m->src); m->src = NULL; m->haveW = 0;
}

/* SubbandDS::_allocBFWeight (:1124-1134) + beamformerWeights::_allocWeights (:896-936): everything zero */
static void alloc_weights(orc_mmi* m, int NC)
{
  free_weights(m);
  const size_t M = m->M, C = m->C, bs = C - NC;
  m->src = (src_t*) calloc(m->nSource, sizeof(src_t)); m->NC = NC;
  for (int s = 0; s < m->nSource; s++) {
    m->src[s].wq = (zc*) calloc(M * C, sizeof(zc)); m->src[s].B = (zc*) calloc(M * C * bs, sizeof(zc)); m->src[s].wa = (zc*) calloc(M * bs, sizeof(zc));
    m->src[s].wl = (zc*) calloc(M * C, sizeof(zc)); m->src[s].ta = (zc*) calloc(M * C, sizeof(zc)); m->src[s].csd = (zc*) calloc(M * C * C, sizeof(zc));
  }
  memset(m->wp1, 0, sizeof(zc) * M);
  m->haveW = 1;
}

orc_mmi* orc_mmi_create(int fftLen, int chanN, int halfBandShift, int targetSourceX, int nSource, int pfType, double alpha)
{
  orc_mmi* m = (orc_mmi*) calloc(1, sizeof(orc_mmi));
  m->M = fftLen; m->C = chanN; m->hbs = halfBandShift; m->target = targetSourceX; m->nSource = nSource; m->pfType = pfType; m->alpha = alpha;
  m->frameX = -1;                                                           /* FrameResetX */
  m->wp1 = (zc*) calloc(fftLen, sizeof(zc)); m->vec = (zc*) calloc(fftLen, sizeof(zc));
  return m;
}

void orc_mmi_free(orc_mmi* m)
{
  if (!m) return;
  free_weights(m);
  if (m->itf) { for (int s = 0; s < m->nSource; s++) free(m->itf[s]); free(m->itf); }
  free(m->avgOut); free(m->wp1); free(m->vec); free(m);
}

void orc_mmi_reset(orc_mmi* m) { m->frameX = -1; }                           /* the densities and the averaged output stay (stream.h reset) */

void orc_mmi_use_binary_mask(orc_mmi* m, double avgFactor, unsigned fwidth, unsigned type)
{
  /* beamformer.h:283-297 */
  m->avgFactor = avgFactor; m->fwidth = fwidth; m->useMask = 1; m->maskType = type;
  if (m->itf) { for (int s = 0; s < m->nSource; s++) free(m->itf[s]); free(m->itf); }
  m->itf = (zc**) calloc(m->nSource, sizeof(zc*));
  for (int s = 0; s < m->nSource; s++) m->itf[s] = (zc*) calloc(m->M, sizeof(zc));
  free(m->avgOut); m->avgOut = (zc*) calloc(m->M, sizeof(zc));
}

/* ---- _calcBlockingMatrix (:398-479), any number of constraints; B: [C][C-NC] */
static int blocking_matrix_nc(const zc* d, int C, int NC, zc* B)
{
  const int bsize = C - NC;
  if (bsize <= 0) return 0;
  zc* P = (zc*) malloc(sizeof(zc) * C * C); zc* vec = (zc*) malloc(sizeof(zc) * C);
  double nrm = 0.0;
  for (int i = 0; i < C; i++) nrm += gabs2(d[i]);
  nrm = sqrt(nrm); nrm = nrm * nrm;
  for (int i = 0; i < C; i++) for (int j = 0; j < C; j++) P[i * C + j] = ((i == j) ? 1.0 : 0.0) + gmul(gmul(-1.0 / nrm, conj(d[i])), d[j]);   /* zgeru */
  memset(B, 0, sizeof(zc) * C * bsize);
  for (int idim = 0; idim < bsize; idim++) {
    for (int i = 0; i < C; i++) vec[i] = P[i * C + idim];
    for (int jdim = 0; jdim < idim; jdim++) {
      zc ip = 0.0;
      for (int i = 0; i < C; i++) ip += gmul(conj(B[i * bsize + jdim]), vec[i]);
      ip = gmulr(ip, -1.0);
      for (int i = 0; i < C; i++) vec[i] += gmul(ip, B[i * bsize + jdim]);
    }
    double nv = 0.0;
    for (int i = 0; i < C; i++) nv += gabs2(vec[i]);
    nv = sqrt(nv);
    for (int i = 0; i < C; i++) B[i * bsize + idim] = gmulr(vec[i], 1.0 / nv);
  }
  free(P); free(vec);
  return 1;
}

/* ---- beamformerWeights::calcMainlobe (:531-594): wq = e^{j val} / C, ta = wq, blocking matrices for every bin when isGSC */
static void calc_mainlobe(orc_mmi* m, src_t* w, double fs, const double* delays, int isGSC)
{
  const int M = m->M, C = m->C, M2 = M / 2;
  if (m->hbs) {
    const float fshift = 0.5f;
    for (int f = 0; f < M2; f++)
      for (int c = 0; c < C; c++) {
        const double val = -2.0 * M_PI * (fshift + f) * fs * delays[c] / M;
        w->wq[(size_t) f * C + c] = gdivr(gpolar(1.0, val), C);
        w->wq[(size_t) (M - 1 - f) * C + c] = gdivr(gpolar(1.0, -val), C);
      }
  } else {
    for (int c = 0; c < C; c++) w->wq[c] = gdivr(gpolar(1.0, 0.0), C);
    for (int f = 1; f < M2; f++)
      for (int c = 0; c < C; c++) {
        const double val = -2.0 * M_PI * f * delays[c] * fs / M;
        w->wq[(size_t) f * C + c] = gdivr(gpolar(1.0, val), C);
        w->wq[(size_t) (M - f) * C + c] = gdivr(gpolar(1.0, -val), C);
      }
    for (int c = 0; c < C; c++) { const double val = -M_PI * fs * delays[c]; w->wq[(size_t) M2 * C + c] = gdivr(gpolar(1.0, val), C); }
  }
  memcpy(w->ta, w->wq, sizeof(zc) * (size_t) M * C);                         /* setTimeAlignment :992-997 */
  if (isGSC) for (int f = 0; f < M; f++) blocking_matrix_nc(&w->wq[(size_t) f * C], C, 1, &w->B[(size_t) f * C * (C - m->NC)]);
}

/* ---- putInverseMat22 (:202-242) */
static void put_inverse_mat22(zc* mat /*[2][2]*/)
{
  const double beta = 0.01;
  zc m00 = mat[0], m11 = mat[3], m01 = mat[1], m10 = mat[2];
  zc det = gmul(m00, m11) - gmul(m01, m10);
  if (hypot(creal(det), cimag(det)) < 1.0E-07) {
    m00 = (creal(m00) + beta) + I * cimag(m00); m11 = (creal(m11) + beta) + I * cimag(m11);
    det = gmul(m00, m11) - gmul(m01, m10);
  }
  mat[0] = gdiv(m11, det); mat[3] = gdiv(m00, det);
  mat[1] = gmulr(gdiv(m01, det), -1.0); mat[2] = gmulr(gdiv(m10, det), -1.0);
}

/* ---- calcNullBeamformer (:314-394): wt <- Cm inv(Cm^H Cm) g, Cm = [wt, pWj...], g = e_0 */
static int calc_null_beamformer(zc* wt, zc** pWj, int C, int NC)
{
  zc* Cm = (zc*) malloc(sizeof(zc) * C * NC); zc* inv = (zc*) calloc((size_t) NC * NC, sizeof(zc)); zc* v = (zc*) calloc(NC, sizeof(zc));
  for (int i = 0; i < C; i++) { Cm[i * NC] = wt[i]; for (int j = 1; j < NC; j++) Cm[i * NC + j] = pWj[j - 1][i]; }
  for (int a = 0; a < NC; a++) for (int b = 0; b < NC; b++) { zc acc = 0.0; for (int i = 0; i < C; i++) acc += gmul(conj(Cm[i * NC + a]), Cm[i * NC + b]); inv[a * NC + b] = acc; }
  if (NC != 2) {
    double* tmp = (double*) malloc(sizeof(double) * 2 * NC * NC); double* out = (double*) malloc(sizeof(double) * 2 * NC * NC);
    for (int k = 0; k < NC * NC; k++) { tmp[2 * k] = creal(inv[k]); tmp[2 * k + 1] = cimag(inv[k]); }
    orc_pseudoinverse(tmp, NC, out, 1.0E-8f);                               /* pseudoinverse(invMat, invMat): the result is taken whatever it returns */
    for (int k = 0; k < NC * NC; k++) inv[k] = out[2 * k] + I * out[2 * k + 1];
    free(tmp); free(out);
  } else put_inverse_mat22(inv);
  for (int a = 0; a < NC; a++) { zc acc = 0.0; for (int b = 0; b < NC; b++) acc += gmul(inv[a * NC + b], (b == 0) ? 1.0 : 0.0); v[a] = acc; }
  for (int i = 0; i < C; i++) { zc acc = 0.0; for (int a = 0; a < NC; a++) acc += gmul(Cm[i * NC + a], v[a]); wt[i] = acc; }
  free(Cm); free(inv); free(v);
  return 1;
}

/* ---- beamformerWeights::calcMainlobeN (:631-753) */
static void calc_mainlobe_n(orc_mmi* m, src_t* w, double fs, const double* delaysT, const double* delaysIs /*[NC-1][C]*/, int NC, int isGSC)
{
  const int M = m->M, C = m->C, M2 = M / 2;
  zc** pWj = (zc**) malloc(sizeof(zc*) * (NC - 1)); zc** pWjConj = (zc**) malloc(sizeof(zc*) * (NC - 1));
  for (int n = 0; n < NC - 1; n++) { pWj[n] = (zc*) calloc(C, sizeof(zc)); pWjConj[n] = (zc*) calloc(C, sizeof(zc)); }
  calc_mainlobe(m, w, fs, delaysT, 0);
  if (m->hbs) {
    const float fshift = 0.5f;
    for (int f = 0; f < M2; f++) {
      zc* vec = &w->wq[(size_t) f * C]; zc* vecConj = &w->wq[(size_t) (M - 1 - f) * C];
      for (int c = 0; c < C; c++) {
        vec[c] = gmulr(vec[c], C); vecConj[c] = gmulr(vecConj[c], C);
        for (int n = 0; n < NC - 1; n++) {
          const double valJ = -2.0 * M_PI * (fshift + f) * fs * delaysIs[n * C + c] / M;
          pWj[n][c] = gpolar(1.0, valJ); pWjConj[n][c] = gpolar(1.0, -valJ);
        }
      }
      calc_null_beamformer(vec, pWj, C, NC); calc_null_beamformer(vecConj, pWjConj, C, NC);
    }
  } else {
    zc* vec = &w->wq[0];
    for (int c = 0; c < C; c++) vec[c] = 1.0 / C;
    for (int f = 1; f < M2; f++) {
      vec = &w->wq[(size_t) f * C];
      for (int c = 0; c < C; c++) {
        vec[c] = gmulr(vec[c], C);
        for (int n = 0; n < NC - 1; n++) { const double valJ = -2.0 * M_PI * f * fs * delaysIs[n * C + c] / M; pWj[n][c] = gpolar(1.0, valJ); }
      }
      calc_null_beamformer(vec, pWj, C, NC);                                 /* (the mirror bins M - f keep their delay-and-sum vectors) */
    }
    /* bin M/2, as shipped (:718-731): every channel's entry is overwritten with an interference phase term and the null beamformer is
       solved inside the channel loop, with the interference manifolds left over from bin M/2 - 1 */
    vec = &w->wq[(size_t) M2 * C];
    for (int c = 0; c < C; c++) {
      vec[c] = gmulr(vec[c], C);
      for (int n = 0; n < NC - 1; n++) { const double val = -M_PI * fs * delaysIs[n * C + c]; vec[c] = gdivr(gpolar(1.0, val), C); }
      calc_null_beamformer(vec, pWj, C, NC);
    }
  }
  if (isGSC) for (int f = 0; f < M; f++) blocking_matrix_nc(&w->wq[(size_t) f * C], C, NC, &w->B[(size_t) f * C * (C - NC)]);
  for (int n = 0; n < NC - 1; n++) { free(pWj[n]); free(pWjConj[n]); }
  free(pWj); free(pWjConj);
}

int orc_mmi_calc_weights(orc_mmi* m, double sampleRate, const double* delays)
{
  /* :1769-1780 */
  alloc_weights(m, 1);
  for (int s = 0; s < m->nSource; s++) calc_mainlobe(m, &m->src[s], sampleRate, delays + (size_t) s * m->C, 1);
  return 0;
}

int orc_mmi_calc_weights_n(orc_mmi* m, double sampleRate, const double* delays, unsigned NC)
{
  /* :1788-1811; the weight objects are only allocated when there are none yet (:1790-1791) */
  if (NC < 2 || NC > (unsigned) m->C || NC > (unsigned) m->nSource) return -1;   /* calcMainlobeN :633-635; rows of the delay matrix */
  if (!m->haveW) alloc_weights(m, (int) NC);
  if ((unsigned) m->NC != NC) return -2;                                     /* (the reference would run over its buffers) */
  double* delaysIs = (double*) malloc(sizeof(double) * (NC - 1) * m->C);
  for (int s = 0; s < m->nSource; s++) {
    for (unsigned srcY = 0, i = 0; i < NC - 1; srcY++) {
      if ((int) srcY == s) continue;
      memcpy(delaysIs + (size_t) i * m->C, delays + (size_t) srcY * m->C, sizeof(double) * m->C); i++;
    }
    calc_mainlobe_n(m, &m->src[s], sampleRate, delays + (size_t) s * m->C, delaysIs, (int) NC, 1);
  }
  free(delaysIs);
  return 0;
}

/* wl = B wa (calcSidelobeCancellerP_f / U_f :761-799) */
static void update_wl(orc_mmi* m, src_t* w, unsigned f)
{
  const int C = m->C, bs = C - m->NC;
  for (int i = 0; i < C; i++) { zc acc = 0.0; for (int j = 0; j < bs; j++) acc += gmul(w->B[((size_t) f * C + i) * bs + j], w->wa[(size_t) f * bs + j]); w->wl[(size_t) f * C + i] = acc; }
}

/* scaling (:1455-1490): W [Msrc][N] <- diag(Wp[stdsnsr][i]) W with Wp the pseudo-inverse (csvdc in complex<float>).  For Msrc < N the
   shipped pseudoinverse reads singular values and left vectors it never computed (:283-297 loop to N); here those terms are zero. */
static void scaling(zc* W, int Ms, int N, float thr)
{
  double* A = (double*) malloc(sizeof(double) * 2 * Ms * N); double* Wp = (double*) calloc((size_t) 2 * N * Ms, sizeof(double));
  for (int k = 0; k < Ms * N; k++) { A[2 * k] = creal(W[k]); A[2 * k + 1] = cimag(W[k]); }
  orc_pseudoinverse_mn(A, Ms, N, Wp, thr);
  const int stdsnsr = N / 2;
  for (int i = 0; i < Ms; i++)
    for (int j = 0; j < N; j++) {
      const zc wp = Wp[2 * ((size_t) stdsnsr * Ms + i)] + I * Wp[2 * ((size_t) stdsnsr * Ms + i) + 1];
      W[i * N + j] = gmul(wp, W[i * N + j]);
    }
  free(A); free(Wp);
}

int orc_mmi_set_active_weights_f(orc_mmi* m, unsigned fbinX, const double* packed, int option)
{
  /* :1821-1880 */
  if (!m->haveW) return -1;
  if (fbinX >= (unsigned) m->M) return -2;
  const int C = m->C, bs = C - m->NC, S = m->nSource;
  for (int s = 0; s < S; s++) {
    for (int c = 0; c < bs; c++) m->src[s].wa[(size_t) fbinX * bs + c] = packed[(size_t) s * 2 * bs + 2 * c] + I * packed[(size_t) s * 2 * bs + 2 * c + 1];
    update_wl(m, &m->src[s], fbinX);
  }
  zc* Wl = (zc*) malloc(sizeof(zc) * S * C);
  for (int s = 0; s < S; s++) for (int c = 0; c < C; c++) Wl[s * C + c] = conj(m->src[s].wl[(size_t) fbinX * C + c]);
  if (option == 1) scaling(Wl, S, C, 1.0E-7f);
  for (int s = 0; s < S; s++) for (int c = 0; c < C; c++) m->src[s].wl[(size_t) fbinX * C + c] = conj(Wl[s * C + c]);   /* setSidelobeCanceller_f */
  free(Wl);
  return 0;
}

int orc_mmi_set_hi_active_weights_f(orc_mmi* m, unsigned fbinX, const double* pkdWa, const double* pkdwb, int option)
{
  /* :1891-1968; pkdWa: [nSrc][C-NC][nSrc] complex, pkdwb: [nSrc][nSrc] complex */
  if (!m->haveW) return -1;
  if (fbinX >= (unsigned) m->M) return -2;
  const int C = m->C, bs = C - m->NC, S = m->nSource;
  zc* Wc = (zc*) malloc(sizeof(zc) * S * S);
  for (int k = 0; k < S * S; k++) Wc[k] = conj(pkdwb[2 * k] + I * pkdwb[2 * k + 1]);
  if (option == 1) scaling(Wc, S, S, 1.0E-7f);
  for (int s = 0; s < S; s++) {
    src_t* w = &m->src[s];
    for (int c = 0; c < bs; c++) {                                           /* we = Wa[s] wb[s] */
      zc acc = 0.0;
      for (int y = 0; y < S; y++) { const size_t i = ((size_t) s * bs + c) * S + y; acc += gmul(pkdWa[2 * i] + I * pkdWa[2 * i + 1], conj(Wc[s * S + y])); }
      w->wa[(size_t) fbinX * bs + c] = acc;
    }
    update_wl(m, w, fbinX);
  }
  free(Wc);
  return 0;
}

void orc_mmi_get(const orc_mmi* m, int srcX, int kind, double* out)
{
  const size_t M = m->M, C = m->C, bs = C - m->NC; const src_t* w = &m->src[srcX];
  const zc* p = kind == 0 ? w->wq : kind == 1 ? w->wl : kind == 2 ? w->B : kind == 3 ? w->ta : w->wa;
  const size_t n = kind == 2 ? M * C * bs : kind == 4 ? M * bs : M * C;
  for (size_t k = 0; k < n; k++) { out[2 * k] = creal(p[k]); out[2 * k + 1] = cimag(p[k]); }
}

/* calcOutputOfGSC (:1251-1287), normalizeWeight = false */
static zc gsc_out(const zc* x, const zc* wl, const zc* wq, int C)
{
  zc acc = 0.0;
  for (int i = 0; i < C; i++) acc += gmul(conj(wq[i] - wl[i]), x[i]);
  return acc;
}
static zc dotc(const zc* w, const zc* x, int C) { zc acc = 0.0; for (int i = 0; i < C; i++) acc += gmul(conj(w[i]), x[i]); return acc; }

/* ZelinskiFilter_f (postfilter.cc:59-136) */
static double zelinski_f(const zc* d, const zc* x, int C, zc* prev, double alpha, int pfType)
{
  zc ta[C];
  for (int i = 0; i < C; i++) ta[i] = gmul(conj(d[i]), x[i]);
  zc sum = 0.0;
  for (int i = 0; i < C - 1; i++)
    for (int j = i + 1; j < C; j++) {
      const int idx = i * C + j;
      const zc xx = gmul(ta[i], conj(ta[j]));
      zc est = xx;
      if (alpha > 0.0) est = gmulr(prev[idx], alpha) + gmulr(xx, 1.0 - alpha);
      sum = sum + est; prev[idx] = est;
    }
  double numerator;
  if (1 & pfType) { numerator = creal(sum); if (numerator < 0.0) numerator = 0.0; } else numerator = hypot(creal(sum), cimag(sum));
  double denominator = 0.0;
  for (int i = 0; i < C; i++) {
    const int idx = i * C + i;
    double est;
    if (alpha > 0.0) est = alpha * creal(prev[idx]) + (1.0 - alpha) * gabs2(ta[i]); else est = gabs2(ta[i]);
    denominator += est; prev[idx] = est;
  }
  double W = (numerator / denominator) * (2.0 / (C - 1.0));
  if (W >= 1.0) W = 1.0;
  if (W < 0.0001) W = 0.0001;
  return W;
}

/* ZelinskiFilter (postfilter.cc:158-221); X: snapshots [C][Fin] of this frame */
static void zelinski_filter(orc_mmi* m, const zc* manifold /*[M][C]*/, const zc* X, int Fin, zc* sig, zc* csd, double alpha, int pfType)
{
  const int M = m->M, C = m->C, M2 = M / 2;
  zc x[C];
  const int last = m->hbs ? M - 1 : M2;
  for (int f = 0; f <= last; f++) {
    for (int c = 0; c < C; c++) x[c] = X[(size_t) c * Fin + f];
    const double r = zelinski_f(&manifold[(size_t) f * C], x, C, &csd[(size_t) f * C * C], alpha, pfType);
    const zc wf = gpolar(r, 0);
    m->wp1[f] = wf;
    if (!m->hbs && f > 0 && f < M2) m->wp1[M - f] = conj(wf);
  }
  if (pfType == 0) return;
  for (int f = 0; f <= last; f++) {
    const zc outf = gmul(m->wp1[f], sig[f]);
    sig[f] = outf;
    if (!m->hbs && f > 0 && f < M2) sig[M - f] = conj(outf);
  }
}

/* ApabFilter_f (postfilter.cc:225-264), channelX >= 0 branch */
static double apab_f(const zc* propagation, const zc* snapShot, int nChan, zc y, int channelX)
{
  (void) nChan;
  const double phi_yy = gabs2(y);
  const zc dsf = conj(propagation[channelX]);
  const zc xsf = snapShot[channelX];
  const zc ysf = gmul(dsf, xsf);
  const double phi_xx = gabs2(ysf);
  double Wf = phi_yy / phi_xx;
  if (Wf >= 1.0) Wf = 1.0;
  if (Wf <= -1.0) Wf = -1.0;
  return Wf;
}

/* ApabFilter (postfilter.cc:275-340) */
static void apab_filter(orc_mmi* m, const zc* manifold /*[M][C]*/, const zc* X, int Fin, zc* sig, int channelX)
{
  const int M = m->M, C = m->C, M2 = M / 2;
  zc x[C]; zc windowV[M];
  if (channelX < 0) channelX = C / 2;
  for (int f = 0; f < M; f++) windowV[f] = 0.0;
  for (int f = 0; f < M2; f++) {
    for (int c = 0; c < C; c++) x[c] = X[(size_t) c * Fin + f];
    const double r = apab_f(&manifold[(size_t) f * C], x, C, sig[f], channelX);
    const zc wf = gpolar(r, 0);
    windowV[f] = wf;
    if (m->hbs) windowV[M - 1 - f] = conj(wf);
  }
  const int length = m->hbs ? M : M2;
  for (int f = 0; f < length; f++) sig[f] = gmul(windowV[f], sig[f]);
}

static void post_filter(orc_mmi* m, int srcX, const zc* X, int Fin, zc* sig)
{
  /* :2036-2060 and :2163-2190 */
  const double alpha = (m->frameX > 0) ? m->alpha : 0.0;
  if (0x04 & m->pfType) apab_filter(m, m->src[srcX].ta, X, Fin, sig, m->C / 2);
  else if (0x01 & m->pfType || 0x02 & m->pfType) {
    const zc* wq = (0x08 & m->pfType) ? m->src[srcX].wq : m->src[srcX].ta;
    if (m->frameX < 0) zelinski_filter(m, wq, X, Fin, sig, m->src[srcX].csd, alpha, 0);        /* MINFRAMES 0 (:1136) */
    else zelinski_filter(m, wq, X, Fin, sig, m->src[srcX].csd, alpha, m->pfType);
  }
}

/* calcInterferenceOutputs (:2079-2194) */
static void interference_outputs(orc_mmi* m, const zc* X, int Fin)
{
  const int M = m->M, C = m->C, M2 = M / 2;
  zc x[C];
  for (int s = 0; s < m->nSource; s++) {
    if (m->maskType == 0 && s == m->target) continue;
    const src_t* w = &m->src[s]; zc* o = m->itf[s];
    if (m->hbs) {
      for (int f = 0; f < M; f++) {
        for (int c = 0; c < C; c++) x[c] = X[(size_t) c * Fin + f];
        o[f] = (m->maskType == 0) ? gsc_out(x, &w->wl[(size_t) f * C], &w->wq[(size_t) f * C], C) : dotc(&w->wq[(size_t) f * C], x, C);
      }
    } else {
      for (int c = 0; c < C; c++) x[c] = X[(size_t) c * Fin];
      o[0] = dotc(&w->wq[0], x, C);
      for (int f = 1; f <= M2; f++) {
        for (int c = 0; c < C; c++) x[c] = X[(size_t) c * Fin + f];
        const zc val = (m->maskType == 0) ? gsc_out(x, &w->wl[(size_t) f * C], &w->wq[(size_t) f * C], C) : dotc(&w->wq[(size_t) f * C], x, C);
        if (f < M2) { o[f] = val; o[M - f] = conj(val); } else o[M2] = val;
      }
    }
  }
  for (int s = 0; s < m->nSource; s++) {
    if (m->maskType == 0 && s == m->target) continue;
    post_filter(m, s, X, Fin, m->itf[s]);
  }
}

/* getMeanOfSubbandC (:2212-2231) */
static zc mean_of_subband(int fbinX, const zc* output, unsigned fftLen, unsigned fwidth)
{
  if (fwidth <= 1) return output[fbinX];
  int fbinStart, fbinEnd; unsigned count = 0; zc sum = 0.0;
  fbinStart = fbinX - fwidth / 2;
  if (fbinStart < 1) fbinStart = 1;
  fbinEnd = fbinX + fwidth / 2;
  if (fbinEnd >= fftLen) fbinEnd = fftLen - 1;
  for (int i = fbinStart; i <= fbinEnd; i++, count++) sum = sum + output[i];
  return gdivr(sum, (double) count);
}

/* binaryMasking (:2241-2319) */
static void binary_masking(orc_mmi* m, zc* output)
{
  const int M = m->M, M2 = M / 2;
  const zc* tgt = m->itf[m->target];
  const int f0 = m->hbs ? 0 : 1, f1 = m->hbs ? M - 1 : M2;
  for (int f = f0; f <= f1; f++) {
    double maxPow = 0.0;
    const double tgtPow = gabs2(tgt[f]);
    for (int s = 0; s < m->nSource; s++) {
      if (s == m->target) continue;
      const double valPow = gabs2(m->itf[s][f]);
      if (valPow > maxPow) maxPow = valPow;
    }
    zc newVal = 0.0;
    if (m->avgFactor >= 0.0) newVal = gmulr(mean_of_subband(f, m->avgOut, m->hbs ? (unsigned) M : (unsigned) (M / 2), m->fwidth), m->avgFactor);
    if (tgtPow < maxPow) {
      if (m->hbs) output[f] = newVal;
      else if (f < M2) { output[f] = newVal; output[M - f] = newVal; }       /* as shipped: the mirror bin gets the value itself, not its conjugate */
      else output[M2] = newVal;
      if (m->avgFactor >= 0.0) m->avgOut[f] = newVal;
    } else if (m->avgFactor >= 0.0) {
      m->avgOut[f] = gmulr(m->avgOut[f], m->avgFactor) + gmulr(output[f], 1.0 - m->avgFactor);   /* setAveragedOutput :2201-2206 */
    }
  }
}

int orc_mmi_next(orc_mmi* m, const double* Xd, int Fin, double* out)
{
  /* :1973-2072; Xd: [C][Fin] complex, Fin = M (halfBandShift) or >= M/2+1; out: [M] complex */
  if (!m->haveW) return -1;
  const int M = m->M, C = m->C, M2 = M / 2;
  const zc* X = (const zc*) Xd;
  const src_t* w = &m->src[m->target];
  zc x[C]; zc* vec = m->vec;
  if (m->hbs) {
    for (int f = 0; f < M; f++) {
      for (int c = 0; c < C; c++) x[c] = X[(size_t) c * Fin + f];
      vec[f] = gsc_out(x, &w->wl[(size_t) f * C], &w->wq[(size_t) f * C], C);
    }
  } else {
    for (int c = 0; c < C; c++) x[c] = X[(size_t) c * Fin];
    vec[0] = dotc(&w->wq[0], x, C);
    for (int f = 1; f <= M2; f++) {
      for (int c = 0; c < C; c++) x[c] = X[(size_t) c * Fin + f];
      const zc val = gsc_out(x, &w->wl[(size_t) f * C], &w->wq[(size_t) f * C], C);
      if (f < M2) { vec[f] = val; vec[M - f] = conj(val); } else vec[M2] = val;
    }
  }
  post_filter(m, m->target, X, Fin, vec);
  if (m->useMask) {
    interference_outputs(m, X, Fin);
    if (m->maskType == 0) memcpy(m->itf[m->target], vec, sizeof(zc) * M);
    binary_masking(m, vec);
  }
  m->frameX++;
  for (int f = 0; f < M; f++) { out[2 * f] = creal(vec[f]); out[2 * f + 1] = cimag(vec[f]); }
  return 0;
}
