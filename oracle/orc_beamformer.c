/*
 * oracle/orc_beamformer.c -- TEST INFRASTRUCTURE (see orc.h).
 * CPU restatement of the subband beamformers:
 *   btk/beamformer/beamformer.cc:531-594   beamformerWeights::calcMainlobe
 *   btk/src/superdirectiveBeamformer.cc:118-137 calcDelaysPolar2
 *   btk/beamformer/beamformer.cc:2486-2553 SubbandMVDR::setDiffuseNoiseModel
 *   btk/beamformer/beamformer.h:362-378    divide(All)NonDiagonalElements
 *   btk/beamformer/beamformer.cc:2555-2581 set(All)Level(s)OfDiagonalLoading
 *   btk/beamformer/beamformer.cc:253-305   pseudoinverse (complex<float> SVD)
 *   btk/beamformer/beamformer.cc:2392-2446 SubbandMVDR::calcMVDRWeights
 *   btk/beamformer/beamformer.cc:61-90,1137-1200,2583-2635 SnapShotArray + DS/MVDR next()
 *   btk/beamformer/beamformer.cc:398-479   _calcBlockingMatrix
 *   btk/beamformer/beamformer.cc:1251-1287,1297-1363 calcOutputOfGSC / SubbandGSC::next
 * The reference's SVD is LINPACK csvdc (in-tree third party, btk/matrix/linpack_c.cc:9518),
 * restated in orc_svd.c and pinned bit for bit against csvdc built from the reference
 * sources (oracle/_ref/, tests/test_oracle_cpu.py).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <complex.h>

typedef double complex zc;
typedef float complex cc;

#define SOUNDSPEED 343740.0   /* superdirectiveBeamformer.cc, mm/s */

static inline zc ZGET(const double* p, size_t i) { return p[2*i] + I * p[2*i+1]; }
static inline void ZSET(double* p, size_t i, zc v) { p[2*i] = creal(v); p[2*i+1] = cimag(v); }

void orc_calc_mainlobe(double fs, const double* delays, int C, int M, double* wq)
{
  /* halfBandShift == false branch, beamformer.cc:557-581 */
  const int M2 = M / 2;
  for (int c = 0; c < C; c++) ZSET(wq, (size_t) 0 * C + c, (cos(0.0) + I * sin(0.0)) / (double) C);
  for (int f = 1; f < M2; f++)
    for (int c = 0; c < C; c++) {
      double val = -2.0 * M_PI * f * delays[c] * fs / M;
      ZSET(wq, (size_t) f * C + c, (cos(val) + I * sin(val)) / (double) C);
      ZSET(wq, (size_t) (M - f) * C + c, (cos(-val) + I * sin(-val)) / (double) C);
    }
  for (int c = 0; c < C; c++) {
    double val = -M_PI * fs * delays[c];
    ZSET(wq, (size_t) M2 * C + c, (cos(val) + I * sin(val)) / (double) C);
  }
}

void orc_calc_mainlobe_hbs(double fs, const double* delays, int C, int M, double* wq)
{
  /* halfBandShift == true branch, beamformer.cc:544-555: bins f and M-1-f, half a bin up (fshift is a float 0.5) */
  const float fshift = 0.5f; const int M2 = M / 2;
  for (int f = 0; f < M2; f++)
    for (int c = 0; c < C; c++) {
      double val = -2.0 * M_PI * (fshift + f) * fs * delays[c] / M;
      ZSET(wq, (size_t) f * C + c, (cos(val) + I * sin(val)) / (double) C);
      ZSET(wq, (size_t) (M - 1 - f) * C + c, (cos(-val) + I * sin(-val)) / (double) C);
    }
}

/* halfBandShift == true: every bin on its own (SubbandDS::next :1159-1175, SubbandGSC::next :1321-1330 through calcOutputOfGSC :1251-1287).
   X [C][T][M], wq [M][C], B [M][C][C-1] and wa [M][C-1] (NULL: delay-and-sum) -> Y [T][M] */
void orc_apply_all_bins(const double* X, const double* wq, const double* B, const double* wa, int C, int T, int M, int normalize, double* Y)
{
  const int bs = C - 1;
  zc* w = (zc*) malloc(sizeof(zc) * (size_t) C);
  for (int f = 0; f < M; f++) {
    double nrm = 0.0;
    for (int i = 0; i < C; i++) {
      zc wl = 0.0;
      if (B) for (int j = 0; j < bs; j++) wl += ZGET(B, ((size_t) f * C + i) * bs + j) * ZGET(wa, (size_t) f * bs + j);
      w[i] = ZGET(wq, (size_t) f * C + i) - wl; nrm += creal(w[i] * conj(w[i]));
    }
    if (B && normalize) { nrm = sqrt(nrm); for (int i = 0; i < C; i++) w[i] = w[i] / (nrm * C); }
    for (int t = 0; t < T; t++) {
      zc val = 0.0;
      for (int c = 0; c < C; c++) val += conj(w[c]) * ZGET(X, ((size_t) c * T + t) * M + f);
      ZSET(Y, (size_t) t * M + f, val);
    }
  }
  free(w);
}

void orc_calc_delays_polar2(float azimuth, float elevation, const double* micpos, int C,
                            double* delays)
{
  /* all arithmetic in float as in the driver (superdirectiveBeamformer.cc:125-134) */
  /* C++ overloads sin(float)/cos(float) -> float versions */
  float c_x = - sinf(elevation) * cosf(azimuth);
  float c_y = - sinf(elevation) * sinf(azimuth);
  float c_z = - cosf(elevation);
  for (int i = 0; i < C; i++) {
    float x = micpos[3*i], y = micpos[3*i+1], z = micpos[3*i+2];
    float t = (c_x * x + c_y * y + c_z * z) / SOUNDSPEED;
    delays[i] = t;
  }
}

static double gsl_sinc(double x)   /* gsl_sf_sinc: sin(pi x)/(pi x) */
{
  double ax = fabs(x);
  if (ax < 1e-300) return 1.0;
  return sin(M_PI * x) / (M_PI * x);
}

void orc_diffuse_noise_model(const double* micpos, int C, int M, double fs, double sspeed, double* R)
{
  double* dm = (double*) calloc((size_t) C * C, sizeof(double));
  for (int m = 0; m < C; m++)
    for (int n = 0; n < m; n++) {
      double dx = micpos[3*m] - micpos[3*n], dy = micpos[3*m+1] - micpos[3*n+1], dz = micpos[3*m+2] - micpos[3*n+2];
      dm[m*C+n] = sqrt(dx*dx + dy*dy + dz*dz);
    }
  for (int f = 0; f <= M / 2; f++) {
    double omega_d_c = 2.0 * fs * f / (M * sspeed);
    double* Rf = R + (size_t) f * C * C * 2;
    for (int m = 0; m < C; m++)
      for (int n = 0; n < m; n++) ZSET(Rf, (size_t) m*C+n, gsl_sinc(omega_d_c * dm[m*C+n]));
    for (int m = 0; m < C; m++) ZSET(Rf, (size_t) m*C+m, 1.0);
    for (int m = 0; m < C; m++)
      for (int n = m + 1; n < C; n++) ZSET(Rf, (size_t) m*C+n, ZGET(Rf, (size_t) n*C+m));
  }
  free(dm);
}

void orc_divide_nondiag(double* R, int C, int M, float mu)
{
  for (int f = 0; f <= M / 2; f++) {
    double* Rf = R + (size_t) f * C * C * 2;
    for (int x = 0; x < C; x++)
      for (int y = 0; y < C; y++)
        if (x != y) ZSET(Rf, (size_t) x*C+y, ZGET(Rf, (size_t) x*C+y) / ((1.0 + mu) + 0.0 * I));
  }
}

void orc_diagonal_loading(double* R, int C, int M, float w)
{
  for (int f = 0; f <= M / 2; f++) {
    double* Rf = R + (size_t) f * C * C * 2;
    for (int c = 0; c < C; c++) Rf[2*((size_t)c*C+c)] += (double) w;
  }
}

/* orc_pseudoinverse (beamformer.cc:253-305) and the LINPACK csvdc it rests on: orc_svd.c */

void orc_mvdr_weights(const double* wq, const double* R, int C, int M, double thr, double* w)
{
  /* beamformer.cc:2392-2446 */
  double* invR = (double*) malloc(sizeof(double) * 2 * C * C);
  zc* tmpH = (zc*) malloc(sizeof(zc) * C);
  for (int c = 0; c < C; c++) ZSET(w, c, 1.0);            /* DC bin: all ones (:2413-2415) */
  for (int f = 1; f <= M / 2; f++) {
    const double* d = wq + (size_t) f * C * 2;
    int ok = orc_pseudoinverse(R + (size_t) f * C * C * 2, C, invR, (float) thr);
    if (!ok) {                                              /* :2425-2427 */
      memset(invR, 0, sizeof(double) * 2 * C * C);
      for (int c = 0; c < C; c++) invR[2*((size_t)c*C+c)] = 1.0;
    }
    for (int i = 0; i < C; i++) {                           /* tmpH = invR^H d (:2430) */
      zc acc = 0.0;
      for (int j = 0; j < C; j++) acc += conj(ZGET(invR, (size_t) j*C+i)) * ZGET(d, j);
      tmpH[i] = acc;
    }
    zc Lambda = 0.0;                                        /* zdotc(tmpH, d) (:2431) */
    for (int i = 0; i < C; i++) Lambda += conj(tmpH[i]) * ZGET(d, i);
    zc norm = Lambda * (double) C;
    {                                                       /* gsl_complex_div(val, norm) (:2440): s = 1/|b|, then scaled products */
      double sN = 1.0 / hypot(creal(norm), cimag(norm)), sbr = sN * creal(norm), sbi = sN * cimag(norm);
      for (int c = 0; c < C; c++) {
        double ar = creal(tmpH[c]), ai = cimag(tmpH[c]);
        w[2 * ((size_t) f * C + c)] = (ar * sbr + ai * sbi) * sN; w[2 * ((size_t) f * C + c) + 1] = (ai * sbr - ar * sbi) * sN;
      }
    }
  }
  free(invR); free(tmpH);
}

void orc_beamform_apply(const double* X, const double* W, int C, int T, int M, double* Y)
{
  /* Y[f] = w_f^H X[f] for f=0..M/2; Y[M-f] = conj(Y[f]) (beamformer.cc:2609-2631) */
  for (int t = 0; t < T; t++) {
    double* y = Y + (size_t) t * M * 2;
    for (int f = 0; f <= M / 2; f++) {
      zc val = 0.0;
      for (int c = 0; c < C; c++)
        val += conj(ZGET(W, (size_t) f*C+c)) * ZGET(X, ((size_t) c * T + t) * M + f);
      ZSET(y, f, val);
      if (f > 0 && f < M / 2) ZSET(y, M - f, conj(val));
    }
  }
}

int orc_blocking_matrix(const double* d, int C, double* B)
{
  /* NC = 1, beamformer.cc:398-479 */
  const int bsize = C - 1;
  if (bsize <= 0) return 0;
  zc* P = (zc*) malloc(sizeof(zc) * C * C);
  zc* vec = (zc*) malloc(sizeof(zc) * C);
  double nrm = 0.0;
  for (int i = 0; i < C; i++) nrm += creal(ZGET(d, i) * conj(ZGET(d, i)));
  nrm = sqrt(nrm); nrm = nrm * nrm;
  for (int i = 0; i < C; i++)
    for (int j = 0; j < C; j++)
      P[i*C+j] = ((i == j) ? 1.0 : 0.0) + (-1.0 / nrm) * conj(ZGET(d, i)) * ZGET(d, j);
  memset(B, 0, sizeof(double) * 2 * C * bsize);
  for (int idim = 0; idim < bsize; idim++) {
    for (int i = 0; i < C; i++) vec[i] = P[i*C+idim];
    for (int jdim = 0; jdim < idim; jdim++) {
      zc ip = 0.0;
      for (int i = 0; i < C; i++) ip += conj(ZGET(B, (size_t) i*bsize+jdim)) * vec[i];
      ip = -ip;
      for (int i = 0; i < C; i++) vec[i] += ip * ZGET(B, (size_t) i*bsize+jdim);
    }
    double nv = 0.0;
    for (int i = 0; i < C; i++) nv += creal(vec[i] * conj(vec[i]));
    nv = sqrt(nv);
    for (int i = 0; i < C; i++) ZSET(B, (size_t) i*bsize+idim, vec[i] * (1.0 / nv));
  }
  free(P); free(vec);
  return 1;
}

void orc_gsc_apply(const double* X, const double* wq, const double* B, const double* wa,
                   int C, int T, int M, int normalize, double* Y)
{
  /* per bin effective weight w = wq - B wa (optionally / (||w|| C)); bin 0 uses wq only
     (beamformer.cc:1335-1342).  B: [M/2+1][C][C-1], wa: [M/2+1][C-1] */
  const int bs = C - 1;
  double* W = (double*) malloc(sizeof(double) * 2 * (size_t)(M/2+1) * C);
  for (int c = 0; c < C; c++) ZSET(W, c, ZGET(wq, c));
  for (int f = 1; f <= M / 2; f++) {
    const double* Bf = B + (size_t) f * C * bs * 2;
    const double* waf = wa + (size_t) f * bs * 2;
    double nrm = 0.0;
    for (int i = 0; i < C; i++) {
      zc wl = 0.0;
      for (int j = 0; j < bs; j++) wl += ZGET(Bf, (size_t) i*bs+j) * ZGET(waf, j);
      zc w = ZGET(wq, (size_t) f*C+i) - wl;
      ZSET(W, (size_t) f*C+i, w);
      nrm += creal(w * conj(w));
    }
    if (normalize) {
      nrm = sqrt(nrm);
      for (int i = 0; i < C; i++) ZSET(W, (size_t) f*C+i, ZGET(W, (size_t) f*C+i) / (nrm * C));
    }
  }
  orc_beamform_apply(X, W, C, T, M, Y);
  free(W);
}

static void cdiv_gsl_b(double ar, double ai, double br, double bi, double* zr, double* zi)
{ /* gsl_complex_div: s = 1/|b|, scaled operands (gsl complex/math.c) */
  const double s = 1.0 / hypot(br, bi);
  const double sbr = s * br, sbi = s * bi;
  *zr = (ar * sbr + ai * sbi) * s; *zi = (ai * sbr - ar * sbi) * s;
}

/* SubbandGSCRLS (beamformer.h:213-262; beamformer.cc:1497-1698): the GSC whose active weight vectors follow a recursive-least-squares update
 * after every frame (_updateActiveWeightVector2 :1627-1698, notation of Van Trees pp. 766-767):
 *   Z = B^H X;  g = (P Z / mu) / (1 + Z^H P Z / mu)  [written as zdotc(P^H Z, Z)];  P <- (P - g (P^H Z)^H) / mu;
 *   wa <- (I - sigma2 P) wa + g conj(Y);  quadratic constraint (CONSTANT_NORM 1: wa <- wa alpha/||wa||; THRESHOLD_LIMITATION 2: the same when
 *   ||wa||^2 >= alpha);  wl = B wa.
 * The frame's output uses the weights BEFORE the update; bin 0 is wq^H X and is never adapted (:1590-1611).  Every utterance starts from
 * P = P0 (initPrecisionMatrix: I / sigma2init, or setPrecisionMatrix) and wa = 0 -- the reference keeps adapting across reset().
 * X: [C][T][M] complex, wq: [M][C] (rows 0..M/2 used), B: [M/2+1][C][C-1], P0: [M/2+1][n][n] (n = C-1), diagW: [M/2+1] -> Y [T][M] (bins above
 * M/2 mirrored), waOut (optional) [M/2+1][n] final active weights. */
void orc_gsc_rls(const double* X, const double* wq, const double* B, const double* P0, const double* diagW, int C, int T, int M, double myu,
                 double alpha, int qctype, int adapt, int normalize, double* Y, double* waOut)
{
  const int n = C - 1, F = M / 2 + 1;
  zc* P = (zc*) malloc(sizeof(zc) * (size_t) F * n * n); zc* wa = (zc*) calloc((size_t) F * n, sizeof(zc));
  zc* Z = (zc*) malloc(sizeof(zc) * n); zc* PH = (zc*) malloc(sizeof(zc) * n); zc* g = (zc*) malloc(sizeof(zc) * n); zc* wn = (zc*) malloc(sizeof(zc) * n);
  zc* w = (zc*) malloc(sizeof(zc) * C);
  for (size_t i = 0; i < (size_t) F * n * n; i++) P[i] = ZGET(P0, i);
  const double rmu = 1.0 / myu;
  for (int t = 0; t < T; t++) {
    for (int f = 0; f < F; f++) {
      const double* Bf = B + (size_t) f * C * n * 2; zc* Pf = P + (size_t) f * n * n; zc* waf = wa + (size_t) f * n;
      zc y = 0.0;
      if (f == 0) { for (int c = 0; c < C; c++) y += conj(ZGET(wq, c)) * ZGET(X, ((size_t) c * T + t) * M); }
      else {
        double nrm = 0.0;
        for (int i = 0; i < C; i++) {
          zc wl = 0.0; for (int j = 0; j < n; j++) wl += ZGET(Bf, (size_t) i * n + j) * waf[j];
          w[i] = ZGET(wq, (size_t) f * C + i) - wl; nrm += creal(w[i] * conj(w[i]));
        }
        if (normalize) { nrm = sqrt(nrm); for (int i = 0; i < C; i++) w[i] = w[i] / (nrm * C); }
        for (int c = 0; c < C; c++) y += conj(w[c]) * ZGET(X, ((size_t) c * T + t) * M + f);
      }
      ZSET(Y, (size_t) t * M + f, y);
      if (f > 0 && f < M / 2) ZSET(Y, (size_t) t * M + (M - f), conj(y));
      if (f == 0 || !adapt) continue;
      for (int j = 0; j < n; j++) { zc a = 0.0; for (int c = 0; c < C; c++) a += conj(ZGET(Bf, (size_t) c * n + j)) * ZGET(X, ((size_t) c * T + t) * M + f); Z[j] = a; }
      for (int j = 0; j < n; j++) { zc a = 0.0; for (int i = 0; i < n; i++) a += conj(Pf[i * n + j]) * Z[i]; PH[j] = a; }
      for (int i = 0; i < n; i++) { zc a = 0.0; for (int j = 0; j < n; j++) a += Pf[i * n + j] * Z[j]; g[i] = a * rmu; }
      zc de = 0.0; for (int j = 0; j < n; j++) de += conj(PH[j]) * Z[j];
      de = de * rmu + 1.0;
      for (int i = 0; i < n; i++) { double qr, qi; cdiv_gsl_b(creal(g[i]), cimag(g[i]), creal(de), cimag(de), &qr, &qi); g[i] = qr + qi * I; }
      for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Pf[i * n + j] = (Pf[i * n + j] - g[i] * conj(PH[j])) * rmu;
      const zc epA = conj(y);
      for (int i = 0; i < n; i++) {
        zc a = 0.0;
        for (int j = 0; j < n; j++) { zc m1 = Pf[i * n + j] * (-diagW[f]); if (i == j) m1 += 1.0; a += m1 * waf[j]; }
        wn[i] = a + g[i] * epA;
      }
      if (qctype == 1 || qctype == 2) {
        double nr = 0.0; for (int i = 0; i < n; i++) nr += creal(wn[i] * conj(wn[i]));
        nr = sqrt(nr);
        if (qctype == 1 || nr * nr >= alpha) for (int i = 0; i < n; i++) wn[i] = wn[i] * (alpha / nr);
      }
      for (int i = 0; i < n; i++) waf[i] = wn[i];
    }
  }
  if (waOut) for (size_t i = 0; i < (size_t) F * n; i++) ZSET(waOut, i, wa[i]);
  free(P); free(wa); free(Z); free(PH); free(g); free(wn); free(w);
}
