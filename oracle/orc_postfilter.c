/*
 * oracle/orc_postfilter.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc.h).
 *
 * CPU restatement of the Zelinski post-filter of btk/postfilter:
 *   calcCSD                     btk/postfilter/postfilter.cc:8-21
 *   TimeAlignment               :30-43
 *   ZelinskiFilter_f            :56-139   (Eq. (4) of the cited paper; spectral floor 1e-4, clamp at 1)
 *   ZelinskiFilter              :157-221  (halfBandShift == false: bins 0..M/2, conjugate mirror)
 *   ZelinskiPostFilter::next    :428-493  (alpha = 0 for the first two frames, minFrames, TYPE_ZELINSKI2 uses wq)
 *   McCowanPostFilter           :568-945  (noise-coherence corrected estimate of the clean-signal PSD; gsl_complex_div restated)
 * Parity unpinned: the reference holds no outputs for these operators; restated from the source text.
 * All arithmetic is fp64 as in the reference (gsl_complex).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* X: [C][T][F] complex double (snapshots, bins 0..M/2), Y: [T][F] complex double (beamformer output), wq: [F][C] complex double
 * (arrayManifold() -- or wq() when type has TYPE_ZELINSKI2 = 8).  out: [T][F] complex double, wp1: [T][F] double (may be NULL).
 * type: 1 Re(.), 2 |.| (the SWIG default), +8 ZELINSKI2.  returns 0, -1 when C <= 1 (jdimension_error, :63-66) */
int orc_zelinski_postfilter(const double* X, const double* Y, const double* wq, int C, int T, int F, double alpha_, int type, int minFrames,
                            double* out, double* wp1)
{
  if (C <= 1) return -1;
  const int NP = C * C;
  double* csd = (double*) calloc((size_t) F * NP * 2, sizeof(double));       /* prevCSDs[fbin][i*C+j] */
  double* ta = (double*) calloc((size_t) C * 2, sizeof(double));
  for (int t = 0; t < T; t++) {
    const int frameX = t - 1;                                                 /* _frameX before _increment() */
    const double alpha = (frameX > 0) ? alpha_ : 0.0;                         /* :463-466 */
    const int pfType = (frameX < minFrames) ? 0 : type;                       /* :471-476, NO_USE_POST_FILTER = 0 */
    for (int f = 0; f < F; f++) {
      double* prev = csd + (size_t) f * NP * 2;
      for (int i = 0; i < C; i++) {                                           /* TimeAlignment: conj(d_i) x_i */
        const double dr = wq[((size_t) f * C + i) * 2], di = -wq[((size_t) f * C + i) * 2 + 1];
        const double xr = X[(((size_t) i * T + t) * F + f) * 2], xi = X[(((size_t) i * T + t) * F + f) * 2 + 1];
        ta[2*i] = dr * xr - di * xi; ta[2*i+1] = dr * xi + di * xr;            /* gsl_complex_mul(dsf, xsf) */
      }
      double sr = 0.0, si = 0.0;
      for (int i = 0; i < C - 1; i++)
        for (int j = i + 1; j < C; j++) {
          const int idx = i * C + j;
          const double ar = ta[2*i], ai = ta[2*i+1], br = ta[2*j], bi = -ta[2*j+1];   /* xi * conj(xj) */
          const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
          double er, ei;
          if (alpha > 0.0) { er = prev[2*idx] * alpha + pr * (1.0 - alpha); ei = prev[2*idx+1] * alpha + pi * (1.0 - alpha); }
          else { er = pr; ei = pi; }
          sr += er; si += ei; prev[2*idx] = er; prev[2*idx+1] = ei;
        }
      double numerator;
      if (1 & pfType) { numerator = sr; if (numerator < 0.0) numerator = 0.0; }
      else numerator = hypot(sr, si);                                         /* gsl_complex_abs */
      double denominator = 0.0;
      for (int i = 0; i < C; i++) {
        const int idx = i * C + i;
        const double a2 = ta[2*i] * ta[2*i] + ta[2*i+1] * ta[2*i+1];          /* gsl_complex_abs2 */
        double est;
        if (alpha > 0.0) est = alpha * prev[2*idx] + (1.0 - alpha) * a2; else est = a2;
        denominator += est; prev[2*idx] = est; prev[2*idx+1] = 0.0;
      }
      double W = (numerator / denominator) * (2.0 / (C - 1.0));
      if (W >= 1.0) W = 1.0;
      if (W < 0.0001) W = 0.0001;
      if (wp1) wp1[(size_t) t * F + f] = W;
      const double yr = Y[((size_t) t * F + f) * 2], yi = Y[((size_t) t * F + f) * 2 + 1];
      if (pfType == 0) { out[((size_t) t * F + f) * 2] = yr; out[((size_t) t * F + f) * 2 + 1] = yi; }
      else { out[((size_t) t * F + f) * 2] = W * yr - 0.0 * yi; out[((size_t) t * F + f) * 2 + 1] = W * yi + 0.0 * yr; }   /* polar(W, 0) * y */
    }
  }
  free(csd); free(ta);
  return 0;
}

/* McCowanPostFilter::setDiffuseNoiseModel (postfilter.cc:568-626): Gamma_mn = sinc(2 fs f d_mn / (M c)), gsl_sf_sinc(x) = sin(pi x)/(pi x).
 * R: [F][C][C] complex double */
void orc_pf_diffuse_noise_model(const double* micPos, int C, int M, double sampleRate, double sspeed, double* R)
{
  const int F = M / 2 + 1;
  for (int f = 0; f < F; f++) {
    const double omega_d_c = 2.0 * sampleRate * f / (M * sspeed);
    double* Rf = R + (size_t) f * C * C * 2;
    for (int m = 0; m < C; m++)
      for (int n = 0; n < m; n++) {
        const double dx = micPos[3*m] - micPos[3*n], dy = micPos[3*m+1] - micPos[3*n+1], dz = micPos[3*m+2] - micPos[3*n+2];
        const double x = omega_d_c * sqrt(dx * dx + dy * dy + dz * dz);
        const double g = (x == 0.0) ? 1.0 : sin(M_PI * x) / (M_PI * x);
        Rf[(m * C + n) * 2] = g; Rf[(m * C + n) * 2 + 1] = 0.0;
      }
    for (int m = 0; m < C; m++) { Rf[(m * C + m) * 2] = 1.0; Rf[(m * C + m) * 2 + 1] = 0.0; }
    for (int m = 0; m < C; m++) for (int n = m + 1; n < C; n++) { Rf[(m * C + n) * 2] = Rf[(n * C + m) * 2]; Rf[(m * C + n) * 2 + 1] = Rf[(n * C + m) * 2 + 1]; }
  }
}

static void cdiv_gsl(double ar, double ai, double br, double bi, double* zr, double* zi)
{ /* gsl_complex_div: s = 1/|b|, scaled operands (gsl complex/math.c) */
  const double s = 1.0 / hypot(br, bi);
  const double sbr = s * br, sbi = s * bi;
  *zr = (ar * sbr + ai * sbi) * s; *zi = (ai * sbr - ar * sbi) * s;
}

/* McCowanPostFilter::{calculateSpectralDensities_f, estimateAverageOfCleanSignalPSD (the non-ORIGINAL_IAIN_PAPER build), PostFiltering, next}
 * (postfilter.cc:706-744,789-826,833-945).  R: [F][C][C] complex double (noise coherence), threshold = _thresholdOfRij. */
int orc_mccowan_postfilter(const double* X, const double* Y, const double* wq, const double* R, int C, int T, int F, double alpha_, int type,
                           int minFrames, double threshold, double* out, double* wp1)
{
  if (C <= 1) return -1;
  const int NP = C * C;
  double* csd = (double*) calloc((size_t) F * NP * 2, sizeof(double));
  double* ta = (double*) calloc((size_t) C * 2, sizeof(double));
  for (int t = 0; t < T; t++) {
    const int frameX = t - 1;
    const double alpha = (frameX > 0) ? alpha_ : 0.0;
    for (int f = 0; f < F; f++) {
      double* prev = csd + (size_t) f * NP * 2; const double* Rf = R + (size_t) f * NP * 2;
      for (int i = 0; i < C; i++) {
        const double dr = wq[((size_t) f * C + i) * 2], di = -wq[((size_t) f * C + i) * 2 + 1];
        const double xr = X[(((size_t) i * T + t) * F + f) * 2], xi = X[(((size_t) i * T + t) * F + f) * 2 + 1];
        ta[2*i] = dr * xr - di * xi; ta[2*i+1] = dr * xi + di * xr;
      }
      for (int i = 0; i < C - 1; i++)
        for (int j = i + 1; j < C; j++) {
          const int idx = i * C + j;
          const double ar = ta[2*i], ai = ta[2*i+1], br = ta[2*j], bi = -ta[2*j+1];
          const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
          if (alpha > 0.0) { prev[2*idx] = prev[2*idx] * alpha + pr * (1.0 - alpha); prev[2*idx+1] = prev[2*idx+1] * alpha + pi * (1.0 - alpha); }
          else { prev[2*idx] = pr; prev[2*idx+1] = pi; }
        }
      double sumOfPSD = 0.0;
      for (int i = 0; i < C; i++) {
        const int idx = i * C + i;
        const double a2 = ta[2*i] * ta[2*i] + ta[2*i+1] * ta[2*i+1];
        double est;
        if (alpha > 0.0) est = alpha * prev[2*idx] + (1.0 - alpha) * a2; else est = a2;
        sumOfPSD += est; prev[2*idx] = est; prev[2*idx+1] = 0.0;
      }
      const double de = sumOfPSD / C;
      double sr = 0.0, si = 0.0;
      for (int i = 0; i < C - 1; i++) {
        const double phi_ii = prev[2 * (i * C + i)];
        for (int j = i + 1; j < C; j++) {
          const double pr = prev[2 * (i * C + j)], pi = prev[2 * (i * C + j) + 1];
          const double phi_jj = prev[2 * (j * C + j)];
          double Rr = Rf[2 * (i * C + j)], Ri = Rf[2 * (i * C + j) + 1];
          if (Rr > threshold && Ri <= 0.0) { Rr = threshold; Ri = 0.0; }
          const double hs = 0.5 * (phi_ii + phi_jj);
          const double nr = pr - Rr * hs, ni = pi - Ri * hs;                   /* phi_ij - R_ij * 0.5 (phi_ii + phi_jj) */
          const double dr = -Rr + 1.0, di = -Ri;                               /* 1 - R_ij */
          double qr, qi; cdiv_gsl(nr, ni, dr, di, &qr, &qi);
          sr += qr; si += qi;
        }
      }
      const double avg = (1 & type) ? sr : hypot(sr, si);
      const double nu = 2.0 * avg / (C * (C - 1));
      double W = nu / de;
      if (W > 1.0) W = 1.0;
      if (W < 0.0001) W = 0.0001;
      if (wp1) wp1[(size_t) t * F + f] = W;
      const double yr = Y[((size_t) t * F + f) * 2], yi = Y[((size_t) t * F + f) * 2 + 1];
      if (frameX >= minFrames) { out[((size_t) t * F + f) * 2] = yr * W; out[((size_t) t * F + f) * 2 + 1] = yi * W; }
      else { out[((size_t) t * F + f) * 2] = yr; out[((size_t) t * F + f) * 2 + 1] = yi; }
    }
  }
  free(csd); free(ta);
  return 0;
}

/* LefkimmiatisPostFilter::{estimateAverageOfNoiseSignalPSD (the non-ORIGINAL_IAIN_PAPER build), PostFiltering, next} (postfilter.cc:1065-1210)
 * on McCowan's density recursions.  lambda: [F] complex, d^H pinv(R_f) d -- the pseudo-inverse itself (beamformer.cc:253-300: LINPACK csvdc in
 * single precision, singular values below minSV dropped) is computed by the caller (oracle.py: numpy SVD), see the note there. */
int orc_lefkimmiatis_postfilter(const double* X, const double* Y, const double* wq, const double* R, const double* lambda, int C, int T, int F, double alpha_, int type,
                                int minFrames, double threshold, int fbinX1, double* out, double* wp1)
{
  if (C <= 1) return -1;
  const int NP = C * C;
  double* csd = (double*) calloc((size_t) F * NP * 2, sizeof(double));
  double* ta = (double*) calloc((size_t) C * 2, sizeof(double));
  for (int t = 0; t < T; t++) {
    const int frameX = t - 1;
    const double alpha = (frameX > 0) ? alpha_ : 0.0;
    for (int f = 0; f < F; f++) {
      double* prev = csd + (size_t) f * NP * 2; const double* Rf = R + (size_t) f * NP * 2;
      for (int i = 0; i < C; i++) {
        const double dr = wq[((size_t) f * C + i) * 2], di = -wq[((size_t) f * C + i) * 2 + 1];
        const double xr = X[(((size_t) i * T + t) * F + f) * 2], xi = X[(((size_t) i * T + t) * F + f) * 2 + 1];
        ta[2*i] = dr * xr - di * xi; ta[2*i+1] = dr * xi + di * xr;
      }
      for (int i = 0; i < C - 1; i++)
        for (int j = i + 1; j < C; j++) {
          const int idx = i * C + j;
          const double ar = ta[2*i], ai = ta[2*i+1], br = ta[2*j], bi = -ta[2*j+1];
          const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
          if (alpha > 0.0) { prev[2*idx] = prev[2*idx] * alpha + pr * (1.0 - alpha); prev[2*idx+1] = prev[2*idx+1] * alpha + pi * (1.0 - alpha); }
          else { prev[2*idx] = pr; prev[2*idx+1] = pi; }
        }
      double sumOfPSD = 0.0;
      for (int i = 0; i < C; i++) {
        const int idx = i * C + i;
        const double a2 = ta[2*i] * ta[2*i] + ta[2*i+1] * ta[2*i+1];
        double est;
        if (alpha > 0.0) est = alpha * prev[2*idx] + (1.0 - alpha) * a2; else est = a2;
        sumOfPSD += est; prev[2*idx] = est; prev[2*idx+1] = 0.0;
      }
      const double de = sumOfPSD / C;
      double sr = 0.0, si = 0.0;
      for (int i = 0; i < C - 1; i++) {
        const double phi_ii = prev[2 * (i * C + i)];
        for (int j = i + 1; j < C; j++) {
          const double pr = prev[2 * (i * C + j)], pi = prev[2 * (i * C + j) + 1];
          const double phi_jj = prev[2 * (j * C + j)];
          double Rr = Rf[2 * (i * C + j)], Ri = Rf[2 * (i * C + j) + 1];
          if (Rr > threshold && Ri <= 0.0) { Rr = threshold; Ri = 0.0; }
          const double hs = 0.5 * (phi_ii + phi_jj);
          const double nr = pr - Rr * hs, ni = pi - Ri * hs;                   /* phi_ij - R_ij * 0.5 (phi_ii + phi_jj) */
          const double dr = -Rr + 1.0, di = -Ri;                               /* 1 - R_ij */
          double qr, qi; cdiv_gsl(nr, ni, dr, di, &qr, &qi);
          sr += qr; si += qi;
        }
      }
      const double avg = (1 & type) ? sr : hypot(sr, si);
      const double phi_ss = 2.0 * avg / (C * (C - 1));                         /* estimateAverageOfCleanSignalPSD (McCowan's) */
      (void) de;
      /* estimateAverageOfNoiseSignalPSD (postfilter.cc:1065-1103): sum over the pairs of (0.5 (phi_ii + phi_jj) - phi_ij) / (1 - R_ij), the
       * coherence clipped on its real part only */
      double vr = 0.0, vi = 0.0;
      for (int i = 0; i < C - 1; i++) {
        const double phi_ii = prev[2 * (i * C + i)];
        for (int j = i + 1; j < C; j++) {
          const double pr = prev[2 * (i * C + j)], pi = prev[2 * (i * C + j) + 1];
          const double phi_jj = prev[2 * (j * C + j)];
          double Rr = Rf[2 * (i * C + j)], Ri = Rf[2 * (i * C + j) + 1];
          if (Rr > threshold) { Rr = threshold; Ri = 0.0; }
          else if (Rr == 1.0) { Rr = 0.99; Ri = 0.0; }
          const double valr = (phi_ii + phi_jj) * 0.5, vali = 0.0 * 0.5;
          const double nr = valr - pr, ni = vali - pi;
          const double dr = -Rr + 1.0, di = -Ri;
          double qr, qi; cdiv_gsl(nr, ni, dr, di, &qr, &qi);
          vr += qr; vi += qi;
        }
      }
      const double avgv = (1 & type) ? vr : hypot(vr, vi);
      const double phi_vv = 2.0 * avgv / (C * (C - 1));
      double W;
      if (f < fbinX1) W = phi_ss / (phi_ss + phi_vv);
      else {
        const double lam = (1 & type) ? lambda[2 * f] : hypot(lambda[2 * f], lambda[2 * f + 1]);   /* d^H pinv(R) d (postfilter.cc:996-1009) */
        const double phi_nn = phi_vv / lam;
        W = phi_ss / (phi_ss + phi_nn);
      }
      if (W > 1.0) W = 1.0;
      if (W < 0.0001) W = 0.0001;
      if (wp1) wp1[(size_t) t * F + f] = W;
      const double yr = Y[((size_t) t * F + f) * 2], yi = Y[((size_t) t * F + f) * 2 + 1];
      if (frameX >= minFrames) { out[((size_t) t * F + f) * 2] = yr * W; out[((size_t) t * F + f) * 2 + 1] = yi * W; }
      else { out[((size_t) t * F + f) * 2] = yr; out[((size_t) t * F + f) * 2 + 1] = yi; }
    }
  }
  free(csd); free(ta);
  return 0;
}
