/*
 * oracle/orc_lpc.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc.h).
 *
 * CPU restatement of the LPC / MVDR spectral-envelope features of btk/feature:
 *   BaseFeature::fftPower            btk/feature/lpc.cc:44-63
 *   WarpFeature::autoCorrelation     btk/feature/lpc.cc:80-139   (warped autocorrelation + Levinson-Durbin, fp32)
 *   BurgFeature::autoCorrelation     btk/feature/lpc.cc:158-207  (Burg lattice, fp32 with fp64 num/den)
 *   MVDRFeature<>::next              btk/feature/lpc.h:134-195
 *   LPCFeature<>::next               btk/feature/lpc.h:291-331
 * Parity unpinned: the reference holds no outputs for these operators; the arithmetic (types, order of
 * operations, the one-bin shift "because of fft", the first dim/2+1 bins of the 2^ceil(log2 dim)-point
 * spectrum) is restated from the source text.  GSL's radix-2 real transform is replaced by a direct
 * fp64 DFT (same mathematical result; rounding differs at the 1e-15 level).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* lpc.cc:32-35: _log2Length = ceil(log(dim)/log(2)), _npoints = 1 << _log2Length */
int orc_lpc_npoints(int dim)
{
  const unsigned l2 = (unsigned) ceil(log((double) dim) / log(2.0));
  return 1 << l2;
}

/* lpc.cc:44-63.  power: in = dim floats (zero padded to npoints), out = npoints/2+1 floats */
void orc_lpc_fft_power(float* power, int dim)
{
  const int N = orc_lpc_npoints(dim), N2 = N / 2;
  double* t = (double*) calloc((size_t) N, sizeof(double));
  float* out = (float*) calloc((size_t) N2 + 1, sizeof(float));
  for (int i = 0; i < dim; i++) t[i] = (double) power[i];
  for (int k = 0; k <= N2; k++) {
    double re = 0.0, im = 0.0;
    for (int n = 0; n < dim; n++) {
      if (t[n] == 0.0) continue;
      const double a = -2.0 * M_PI * (double) (((long) n * k) % N) / (double) N;
      re += t[n] * cos(a); im += t[n] * sin(a);
    }
    if (k == 0 || k == N2) out[k] = (float) (re * re);                 /* purely real bins of a real transform */
    else out[k] = (float) (re * re + im * im);
  }
  memcpy(power, out, sizeof(float) * ((size_t) N2 + 1));
  free(t); free(out);
}

/* lpc.cc:80-139 */
void orc_lpc_warp_autocorr(const float* X, int dim, int order, float warp, float* LP, float* E)
{
  float* K = (float*) calloc((size_t) order + 1, sizeof(float));
  float* R = (float*) calloc((size_t) order + 1, sizeof(float));
  float* WX = (float*) calloc((size_t) dim + 1, sizeof(float));
  float* WT = (float*) calloc((size_t) dim + 1, sizeof(float));
  const int n1 = order + 1;
  float* A = (float*) calloc((size_t) n1 * n1, sizeof(float));        /* _tmpA(row j, col i) = A[j*n1 + i] */
  float sum = 0.0f;
  for (int i = 0; i < dim; i++) sum += X[i] * X[i];
  R[0] = sum;
  for (int j = 0; j < dim; j++) WX[j] = X[j];
  for (int i = 1; i <= order; i++) {
    for (int j = 0; j < dim; j++) WT[j] = WX[j];
    WX[0] = -warp * WT[0];
    for (int j = 1; j < dim; j++) WX[j] = warp * (WX[j - 1] - WT[j]) + WT[j - 1];
    sum = 0.0f;
    for (int j = 0; j < dim; j++) sum += X[j] * WX[j];
    R[i] = sum;
  }
  E[0] = R[0];
  A[1 * n1 + 0] = 1.0f;
  for (int i = 1; i <= order; i++) {
    K[i] = R[i];
    for (int j = 1; j < i; j++) K[i] -= A[j * n1 + (i - 1)] * R[i - j];
    if (E[i - 1] != 0) K[i] /= E[i - 1]; else K[i] = 1000000000;
    A[i * n1 + i] = K[i];
    for (int j = 1; j <= i - 1; j++) {
      const double val = A[j * n1 + (i - 1)] - K[i] * A[(i - j) * n1 + (i - 1)];     /* float expression, lpc.cc:125-126 */
      A[j * n1 + i] = (float) val;
    }
    E[i] = (1 - K[i] * K[i]) * E[i - 1];
  }
  LP[0] = 1.0f;
  for (int i = 1; i <= order; i++) LP[i] = -A[i * n1 + order];
  free(K); free(R); free(WX); free(WT); free(A);
}

/* lpc.cc:158-207 */
void orc_lpc_burg_autocorr(const float* X, int dim, int order, float* A, float* E)
{
  float* EF = (float*) calloc((size_t) dim, sizeof(float)); float* EB = (float*) calloc((size_t) dim, sizeof(float));
  float* EFP = (float*) calloc((size_t) dim, sizeof(float)); float* EBP = (float*) calloc((size_t) dim, sizeof(float));
  float* Af = (float*) calloc((size_t) order + 1, sizeof(float)); float* K = (float*) calloc((size_t) order + 1, sizeof(float));
  E[0] = 0.0f;
  for (int i = 0; i < dim; i++) E[0] += X[i] * X[i];
  for (int i = 0; i <= order; i++) { Af[i] = 0.0f; A[i] = 0.0f; }
  for (int i = 0; i < dim; i++) { EF[i] = X[i]; EB[i] = X[i]; }
  for (int i = 0; i < order; i++) {
    for (int j = 0; j < dim - i - 1; j++) { EFP[j] = EF[j + 1]; EBP[j] = EB[j]; }
    double num = 0.0, den = 0.0;
    for (int j = 0; j < dim - i - 1; j++) {
      num -= 2 * EBP[j] * EFP[j];
      den += EFP[j] * EFP[j] + EBP[j] * EBP[j];
    }
    K[i] = (float) num / den;                                            /* ((float) num) / den, lpc.cc:186 */
    for (int j = 0; j < dim - i - 1; j++) { EF[j] = EFP[j] + K[i] * EBP[j]; EB[j] = EBP[j] + K[i] * EFP[j]; }
    A[0] = 1.0f;
    for (int j = 0; j <= i + 1; j++) Af[j] = A[i - j + 1];
    for (int j = 1; j <= i + 1; j++) A[j] += K[i] * Af[j];
  }
  free(EF); free(EB); free(EFP); free(EBP); free(Af); free(K);
}

/* MVDRFeature<>::next (lpc.h:134-195, kind 0) and LPCFeature<>::next (lpc.h:291-331, kind 1);
 * method 0 = WarpFeature, 1 = BurgFeature.  frames [T][dim] float -> out [T][dim/2+1] double.
 * returns 0, or -1 when order >= dim/2+1 (the constructor's jparameter_error) */
int orc_lpc_feature(const float* frames, long T, int dim, int order, float warp, int method, int kind, double* out)
{
  if (order >= dim / 2 + 1) return -1;
  const int outN = dim / 2 + 1, tempOrder = 2 * order + 1, N2 = orc_lpc_npoints(dim) / 2;
  float* A = (float*) calloc((size_t) order + 1, sizeof(float)); float* E = (float*) calloc((size_t) order + 1, sizeof(float));
  float* PC = (float*) calloc((size_t) tempOrder, sizeof(float));
  size_t paN = (size_t) dim + 1; if (paN < (size_t) N2 + 1) paN = (size_t) N2 + 1;
  float* PA = (float*) calloc(paN, sizeof(float));
  for (long t = 0; t < T; t++) {
    const float* X = frames + t * dim;
    if (method == 0) orc_lpc_warp_autocorr(X, dim, order, warp, A, E); else orc_lpc_burg_autocorr(X, dim, order, A, E);
    if (kind == 0) {
      for (int i = 0; i <= order; i++) {
        double temp = 0;
        for (int ii = 0; ii <= order - i; ii++) temp += (float) (order + 1 - i - 2 * ii) * A[ii] * A[ii + i];
        if (E[0] > 0) PC[order + i] = (float) -temp; else PC[order + i] = 10000000;
      }
      for (int i = 1; i <= order; i++) PC[order - i] = PC[order + i];
      PA[0] = 0;
      for (int i = 1; i <= tempOrder; i++) PA[i] = PC[i - 1];
      for (int i = tempOrder + 1; i <= dim; i++) PA[i] = 0;
      orc_lpc_fft_power(PA, dim);
      for (int i = 0; i <= dim / 2; i++) {
        double temp = sqrt(PA[i]);
        if (temp > 0) temp = E[0] / temp; else temp = 10000000;
        out[t * outN + i] = temp;
      }
    } else {
      PA[0] = 0;
      for (int i = 1; i <= order + 1; i++) PA[i] = A[i - 1];
      for (int i = order + 2; i <= dim; i++) PA[i] = 0;
      orc_lpc_fft_power(PA, dim);
      for (int i = 0; i <= dim / 2; i++) {
        double temp = PA[i];
        if (temp > 0) temp = (2 * E[0]) / (temp * dim); else temp = 10000000;
        out[t * outN + i] = temp;
      }
    }
  }
  free(A); free(E); free(PC); free(PA);
  return 0;
}
