/*
 * oracle/orc_wpe.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc.h).
 *
 * CPU restatement of SingleChannelWPEDereverberationFeature (btk/dereverberation/dereverberation.cc:28-300):
 *   _getLags :61-72, _calculateRr :93-152, _calculateThetan :156-182 (floor 1e-3), _loadR :184-197, _estimateGn :199-226,
 *   next :228-255, _setBandWidthN :257-265.  The prediction filters start from zero (a fresh object / nextSpeaker(), :283-290);
 *   the reference's reset() keeps them from one utterance to the next.
 * GSL's gsl_linalg_complex_cholesky_decomp / _solve are restated (left-looking Cholesky of the lower triangle, two triangular solves).
 * Parity unpinned: no outputs of this operator ship with the reference.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static unsigned wpe_band(double bandWidth, double sampleRate, int size)
{ if (bandWidth == 0.0) return (unsigned) (size / 2); return (unsigned) ((bandWidth / (sampleRate / 2.0)) * (size / 2)); }

/* in-place Cholesky of the lower triangle of the P x P Hermitian matrix A (interleaved complex, row major); 0 ok, -1 not positive definite */
static int chol_lower(double* A, int P)
{
  for (int j = 0; j < P; j++) {
    double ajj = A[2 * (j * P + j)];
    for (int k = 0; k < j; k++) ajj -= A[2 * (j * P + k)] * A[2 * (j * P + k)] + A[2 * (j * P + k) + 1] * A[2 * (j * P + k) + 1];
    if (ajj <= 0.0) return -1;
    ajj = sqrt(ajj); A[2 * (j * P + j)] = ajj; A[2 * (j * P + j) + 1] = 0.0;
    for (int i = j + 1; i < P; i++) {
      double sr = A[2 * (i * P + j)], si = A[2 * (i * P + j) + 1];
      for (int k = 0; k < j; k++) {                                   /* A_ij -= A_ik conj(A_jk) */
        const double ar = A[2 * (i * P + k)], ai = A[2 * (i * P + k) + 1], br = A[2 * (j * P + k)], bi = -A[2 * (j * P + k) + 1];
        sr -= ar * br - ai * bi; si -= ar * bi + ai * br;
      }
      A[2 * (i * P + j)] = sr / ajj; A[2 * (i * P + j) + 1] = si / ajj;
    }
  }
  return 0;
}
static void chol_solve(const double* L, int P, const double* b, double* x)
{
  for (int i = 0; i < P; i++) {                                       /* L c = b */
    double sr = b[2*i], si = b[2*i+1];
    for (int k = 0; k < i; k++) { const double ar = L[2 * (i * P + k)], ai = L[2 * (i * P + k) + 1]; sr -= ar * x[2*k] - ai * x[2*k+1]; si -= ar * x[2*k+1] + ai * x[2*k]; }
    const double d = L[2 * (i * P + i)]; x[2*i] = sr / d; x[2*i+1] = si / d;
  }
  for (int i = P - 1; i >= 0; i--) {                                  /* L^H x = c */
    double sr = x[2*i], si = x[2*i+1];
    for (int k = i + 1; k < P; k++) { const double ar = L[2 * (k * P + i)], ai = -L[2 * (k * P + i) + 1]; sr -= ar * x[2*k] - ai * x[2*k+1]; si -= ar * x[2*k+1] + ai * x[2*k]; }
    const double d = L[2 * (i * P + i)]; x[2*i] = sr / d; x[2*i+1] = si / d;
  }
}

/* Y: [N][M] complex double -> out [N][M]; gn (optional) [M][P].  returns 0, -1 Cholesky failed (GSL would abort), -2 bad parameters */
int orc_wpe_single_w(const double* Y, int N, int M, int lowerN, int upperN, int iterationsN, double loadDb, double bandWidth, double sampleRate,
                     const double* gnInit, double* out, double* gnOut);
int orc_wpe_single(const double* Y, int N, int M, int lowerN, int upperN, int iterationsN, double loadDb, double bandWidth, double sampleRate,
                   double* out, double* gnOut)
{ return orc_wpe_single_w(Y, N, M, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate, NULL, out, gnOut); }
/* gnInit (optional, [M][P]): the prediction filters the object still holds from the utterance before -- reset() keeps _gn, only nextSpeaker()
   zeroes it (dereverberation.cc:258-277), so the first theta_n of the next utterance is computed with the previous filters */
int orc_wpe_single_w(const double* Y, int N, int M, int lowerN, int upperN, int iterationsN, double loadDb, double bandWidth, double sampleRate,
                     const double* gnInit, double* out, double* gnOut)
{
  if (upperN < lowerN || N <= 0) return -2;
  if (bandWidth > sampleRate / 2.0) return -2;
  const int P = upperN - lowerN + 1;
  const double loadFactor = pow(10.0, loadDb / 10.0);
  const unsigned lowerBW = wpe_band(bandWidth, sampleRate, M), upperBW = (unsigned) M - lowerBW;
  double* gn = (double*) calloc((size_t) M * P * 2, sizeof(double));
  if (gnInit) memcpy(gn, gnInit, sizeof(double) * (size_t) M * P * 2);
  double* theta = (double*) calloc((size_t) N * M, sizeof(double));
  double* R = (double*) calloc((size_t) P * P * 2, sizeof(double)); double* r = (double*) calloc((size_t) P * 2, sizeof(double));
  double* lag = (double*) calloc((size_t) P * 2, sizeof(double));
  int rc = 0;
#define GETLAGS(b, s) do { for (int l_ = 0; l_ < P; l_++) { const int ix_ = (s) - l_; if (ix_ < 0) { lag[2*l_] = 0.0; lag[2*l_+1] = 0.0; } \
      else { lag[2*l_] = Y[((size_t) ix_ * M + (b)) * 2]; lag[2*l_+1] = Y[((size_t) ix_ * M + (b)) * 2 + 1]; } } } while (0)
  for (int it = 0; it < iterationsN && rc == 0; it++) {
    for (int n = 0; n < N; n++)                                       /* _calculateThetan */
      for (int b = 0; b < M; b++) {
        double cr = Y[((size_t) n * M + b) * 2], ci = Y[((size_t) n * M + b) * 2 + 1];
        if (n >= lowerN) {
          GETLAGS(b, n - lowerN);
          double dr = 0.0, di = 0.0;                                  /* zdotc(gn, lags) = sum conj(gn_l) lag_l */
          for (int l = 0; l < P; l++) { const double gr = gn[((size_t) b * P + l) * 2], gi = -gn[((size_t) b * P + l) * 2 + 1]; dr += gr * lag[2*l] - gi * lag[2*l+1]; di += gr * lag[2*l+1] + gi * lag[2*l]; }
          cr -= dr; ci -= di;
        }
        double th = hypot(cr, ci); if (th < 1.0E-03) th = 1.0E-03;
        theta[(size_t) n * M + b] = th * th;
      }
    for (int b = 0; b < M && rc == 0; b++) {
      if (((unsigned) b > lowerBW) && ((unsigned) b < upperBW)) continue;
      memset(R, 0, sizeof(double) * P * P * 2); memset(r, 0, sizeof(double) * P * 2);
      for (int n = lowerN; n < N; n++) {                              /* _calculateRr */
        const double th = theta[(size_t) n * M + b];
        GETLAGS(b, n - lowerN);
        for (int row = 0; row < P; row++)
          for (int col = 0; col <= row; col++) {
            const double ar = lag[2*row], ai = lag[2*row+1], br = lag[2*col], bi = -lag[2*col+1];
            R[2 * (row * P + col)] += (ar * br - ai * bi) / th; R[2 * (row * P + col) + 1] += (ar * bi + ai * br) / th;
          }
      }
      for (int n = lowerN; n < N; n++) {
        const double th = theta[(size_t) n * M + b];
        const double cr = Y[((size_t) n * M + b) * 2], ci = -Y[((size_t) n * M + b) * 2 + 1];      /* conj(current) */
        GETLAGS(b, n - lowerN);
        for (int l = 0; l < P; l++) { r[2*l] += (cr * lag[2*l] - ci * lag[2*l+1]) / th; r[2*l+1] += (cr * lag[2*l+1] + ci * lag[2*l]) / th; }
      }
      double maxd = 0.0;                                              /* _loadR */
      for (int c = 0; c < P; c++) { const double d = hypot(R[2 * (c * P + c)], R[2 * (c * P + c) + 1]); if (d > maxd) maxd = d; }
      for (int c = 0; c < P; c++) { const double d = hypot(R[2 * (c * P + c)], R[2 * (c * P + c) + 1]) + maxd * loadFactor; R[2 * (c * P + c)] = d; R[2 * (c * P + c) + 1] = 0.0; }
      if (chol_lower(R, P)) { rc = -1; break; }
      chol_solve(R, P, r, gn + (size_t) b * P * 2);
    }
  }
  for (int n = 0; n < N && rc == 0; n++)                              /* next() */
    for (int b = 0; b < M; b++) {
      double cr = Y[((size_t) n * M + b) * 2], ci = Y[((size_t) n * M + b) * 2 + 1];
      if (n >= lowerN && ((unsigned) b <= lowerBW || (unsigned) b >= upperBW)) {
        GETLAGS(b, n - lowerN);
        double dr = 0.0, di = 0.0;
        for (int l = 0; l < P; l++) { const double gr = gn[((size_t) b * P + l) * 2], gi = -gn[((size_t) b * P + l) * 2 + 1]; dr += gr * lag[2*l] - gi * lag[2*l+1]; di += gr * lag[2*l+1] + gi * lag[2*l]; }
        cr -= dr; ci -= di;
      }
      out[((size_t) n * M + b) * 2] = cr; out[((size_t) n * M + b) * 2 + 1] = ci;
    }
#undef GETLAGS
  if (gnOut && rc == 0) memcpy(gnOut, gn, sizeof(double) * (size_t) M * P * 2);
  free(gn); free(theta); free(R); free(r); free(lag);
  return rc;
}

/* MultiChannelWPEDereverberation (dereverberation.cc:281-586): the lag vector stacks the channels ([channel][lag], _getLags :422-437); every
 * channel has its own theta_n (:499-527), weighted correlation matrix/vector over the SAME stacked lags (:439-495), loading (:529-544) and
 * prediction filter (:546-573).  getOutput (:365-395) subtracts, for ALL channels of a frame, the prediction made with the filter of the
 * channel whose feature asked for the frame first (it indexes _Gn with the argument channelX inside the loop over chanX): filterChan >= 0
 * restates that (all channels through Gn[filterChan]); filterChan < 0 gives every channel its own filter.
 * Y: [C][N][M] complex double -> out [C][N][M]; gnOut (optional) [C][M][C*P].  Parity unpinned as for the single-channel operator. */
int orc_wpe_multi(const double* Y, int C, int N, int M, int lowerN, int upperN, int iterationsN, double loadDb, double bandWidth, double sampleRate,
                  int filterChan, double* out, double* gnOut)
{
  if (upperN < lowerN || N <= 0 || C <= 0 || filterChan >= C) return -2;
  if (bandWidth > sampleRate / 2.0) return -2;
  const int P = upperN - lowerN + 1, PT = P * C;
  const double loadFactor = pow(10.0, loadDb / 10.0);
  const unsigned lowerBW = wpe_band(bandWidth, sampleRate, M), upperBW = (unsigned) M - lowerBW;
  double* gn = (double*) calloc((size_t) C * M * PT * 2, sizeof(double));
  double* theta = (double*) calloc((size_t) C * N * M, sizeof(double));
  double* R = (double*) calloc((size_t) PT * PT * 2, sizeof(double)); double* r = (double*) calloc((size_t) PT * 2, sizeof(double));
  double* lag = (double*) calloc((size_t) PT * 2, sizeof(double));
  int rc = 0;
#define YV(c, n, b) (Y + ((((size_t) (c) * N + (n)) * M + (b)) * 2))
#define GETLAGS(b, s) do { int t_ = 0; for (int c_ = 0; c_ < C; c_++) for (int l_ = 0; l_ < P; l_++, t_++) { const int ix_ = (s) - l_; \
      if (ix_ < 0) { lag[2*t_] = 0.0; lag[2*t_+1] = 0.0; } else { lag[2*t_] = YV(c_, ix_, b)[0]; lag[2*t_+1] = YV(c_, ix_, b)[1]; } } } while (0)
#define PREDICT(c, b, dr, di) do { dr = 0.0; di = 0.0; for (int t_ = 0; t_ < PT; t_++) { const double gr = gn[(((size_t) (c) * M + (b)) * PT + t_) * 2], gi = -gn[(((size_t) (c) * M + (b)) * PT + t_) * 2 + 1]; \
      dr += gr * lag[2*t_] - gi * lag[2*t_+1]; di += gr * lag[2*t_+1] + gi * lag[2*t_]; } } while (0)
  for (int it = 0; it < iterationsN && rc == 0; it++) {
    for (int n = 0; n < N; n++)                                       /* _calculateThetan */
      for (int c = 0; c < C; c++)
        for (int b = 0; b < M; b++) {
          double cr = YV(c, n, b)[0], ci = YV(c, n, b)[1];
          if (n >= lowerN) { GETLAGS(b, n - lowerN); double dr, di; PREDICT(c, b, dr, di); cr -= dr; ci -= di; }
          double th = hypot(cr, ci); if (th < 1.0E-03) th = 1.0E-03;
          theta[((size_t) c * N + n) * M + b] = th * th;
        }
    for (int b = 0; b < M && rc == 0; b++) {
      if (((unsigned) b > lowerBW) && ((unsigned) b < upperBW)) continue;
      for (int c = 0; c < C && rc == 0; c++) {
        memset(R, 0, sizeof(double) * PT * PT * 2); memset(r, 0, sizeof(double) * PT * 2);
        for (int n = lowerN; n < N; n++) {                            /* _calculateRr */
          const double th = theta[((size_t) c * N + n) * M + b];
          GETLAGS(b, n - lowerN);
          for (int row = 0; row < PT; row++)
            for (int col = 0; col <= row; col++) {
              const double ar = lag[2*row], ai = lag[2*row+1], br = lag[2*col], bi = -lag[2*col+1];
              R[2 * (row * PT + col)] += (ar * br - ai * bi) / th; R[2 * (row * PT + col) + 1] += (ar * bi + ai * br) / th;
            }
          const double cr = YV(c, n, b)[0], ci = -YV(c, n, b)[1];     /* conj(current) */
          for (int l = 0; l < PT; l++) { r[2*l] += (cr * lag[2*l] - ci * lag[2*l+1]) / th; r[2*l+1] += (cr * lag[2*l+1] + ci * lag[2*l]) / th; }
        }
        double maxd = 0.0;                                            /* _loadR */
        for (int k = 0; k < PT; k++) { const double d = hypot(R[2 * (k * PT + k)], R[2 * (k * PT + k) + 1]); if (d > maxd) maxd = d; }
        for (int k = 0; k < PT; k++) { const double d = hypot(R[2 * (k * PT + k)], R[2 * (k * PT + k) + 1]) + maxd * loadFactor; R[2 * (k * PT + k)] = d; R[2 * (k * PT + k) + 1] = 0.0; }
        if (chol_lower(R, PT)) { rc = -1; break; }
        chol_solve(R, PT, r, gn + ((size_t) c * M + b) * PT * 2);
      }
    }
  }
  for (int n = 0; n < N && rc == 0; n++)                              /* getOutput */
    for (int c = 0; c < C; c++)
      for (int b = 0; b < M; b++) {
        double cr = YV(c, n, b)[0], ci = YV(c, n, b)[1];
        if (n >= lowerN && ((unsigned) b <= lowerBW || (unsigned) b >= upperBW)) {
          GETLAGS(b, n - lowerN);
          double dr, di; const int fc = filterChan >= 0 ? filterChan : c; PREDICT(fc, b, dr, di);
          cr -= dr; ci -= di;
        }
        out[(((size_t) c * N + n) * M + b) * 2] = cr; out[(((size_t) c * N + n) * M + b) * 2 + 1] = ci;
      }
#undef YV
#undef GETLAGS
#undef PREDICT
  if (gnOut && rc == 0) memcpy(gnOut, gn, sizeof(double) * (size_t) C * M * PT * 2);
  free(gn); free(theta); free(R); free(r); free(lag);
  return rc;
}
