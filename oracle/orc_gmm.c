/*
 * oracle/orc_gmm.c -- TEST INFRASTRUCTURE (see orc.h).
 * CPU restatement of diagonal-covariance GMM scoring:
 *   asr/gaussian/codebookBasic.cc:431-554  CodebookBasic::_scoreOpt (nearest Gaussian,
 *       fp32 accumulation d ascending, 4-way unrolled early exit, strict '<' argmin)
 *   asr/gaussian/codebookBasic.cc:645-766  CodebookBasic::_scoreAll (fp64 per-Gaussian,
 *       log-sum with -100 exponent floor)
 *   asr/gaussian/distribBasic.h:110-114    DistribBasic::_score (val = -log w)
 *   asr/gaussian/codebookBasic.cc:557-609  CodebookBasic::logLhood (the "simplified" nearest-Gaussian score:
 *       the same search, finished as 0.5 * min [+ val[argmin]] -- two float roundings, no codebook scale)
 * Compile with -ffp-contract=off: the reference's x86-64 build has no FMA contraction.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

void orc_gmm_score_opt(const orc_cbset* cb, const float* val, const float* x, int T,
                       float* score, int32_t* argmin)
{
  const int D = cb->dimN, D4 = D / 4;
  for (int t = 0; t < T; t++) {
    const float* pattern = x + (size_t) t * D;
    for (int k = 0; k < cb->K; k++) {
      float minDistSum = 1E20; int minDistIdx = 0;
      for (int i = 0; i < cb->refN[k]; i++) {
        const int g = cb->off[k] + i;
        float distSum = cb->pi[k] + cb->det[g];
        const float* pt = pattern; const float* rv = cb->mean + (size_t) g * D; const float* cv = cb->ivar + (size_t) g * D;
        int dimX;
        for (dimX = 0; dimX < D4; dimX++) {
          float diff0;
          if (distSum > minDistSum) break;
          diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++);
          diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++);
          diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++);
          diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++);
        }
        if (dimX == D4)
          for (dimX = 4 * D4; dimX < D; dimX++) { float diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++); }
        if (distSum < minDistSum) { minDistSum = distSum; minDistIdx = i; }
      }
      float sc = 0.5 * (minDistSum + 2 * val[cb->off[k] + minDistIdx]);
      if (cb->scale[k] != 1.0) sc *= cb->scale[k];
      score[(size_t) t * cb->K + k] = sc;
      if (argmin) argmin[(size_t) t * cb->K + k] = minDistIdx;
    }
  }
}

/* CodebookBasic::logLhood(frame, val) (:557-609) for every (frame, codebook); val may be NULL */
void orc_gmm_log_lhood(const orc_cbset* cb, const float* val, const float* x, int T, float* score, int32_t* argmin)
{
  const int D = cb->dimN, dimN_4 = D / 4;
  for (int t = 0; t < T; t++) {
    const float* pattern = x + (size_t) t * D;
    for (int k = 0; k < cb->K; k++) {
      int minDistIdx = 0; float minDistSum = 1.0E20;
      for (int refX = 0; refX < cb->refN[k]; refX++) {
        const int g = cb->off[k] + refX;
        float distSum = cb->pi[k] + cb->det[g];
        const float* pt = pattern; const float* rv = cb->mean + (size_t) g * D; const float* cv = cb->ivar + (size_t) g * D;
        int dimX;
        for (dimX = 0; dimX < dimN_4; dimX++) {
          float diff0;
          if (distSum > minDistSum) break;
          diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++);
          diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++);
          diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++);
          diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++);
        }
        if (dimX == dimN_4)
          for (dimX = 4 * dimN_4; dimX < D; dimX++) { float diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++); }
        if (distSum < minDistSum) { minDistSum = distSum; minDistIdx = refX; }
      }
      minDistSum *= 0.5;                                 /* "assume a fully continuous system" */
      if (val != NULL) minDistSum += val[cb->off[k] + minDistIdx];
      score[(size_t) t * cb->K + k] = minDistSum;
      if (argmin) argmin[(size_t) t * cb->K + k] = minDistIdx;
    }
  }
}

void orc_gmm_score_all(const orc_cbset* cb, const float* val, const float* x, int T, float* score)
{
  const int D = cb->dimN;
  float* logdist = (float*) malloc(sizeof(float) * 257);
  for (int t = 0; t < T; t++) {
    const float* pattern = x + (size_t) t * D;
    for (int k = 0; k < cb->K; k++) {
      const int R = cb->refN[k]; const float* v = val + cb->off[k]; const float sc = cb->scale[k];
      double minlogdist = 1E20;
      for (int i = 0; i < R; i++) {                    /* COV_DIAGONAL :710-727 */
        const int g = cb->off[k] + i;
        const float* pt = pattern; const float* rv = cb->mean + (size_t) g * D; const float* cv = cb->ivar + (size_t) g * D;
        double distSum = cb->pi[k] + cb->det[g];       /* float + float, then to double */
        for (int d = 0; d < D; d++) { double diff0 = *rv++ - *pt++; distSum += diff0 * diff0 * (*cv++); }
        logdist[i] = 0.5 * distSum;
        if (logdist[i] < minlogdist) minlogdist = logdist[i];
      }
      logdist[R] = minlogdist;                          /* float store of the cache (:733) */
      float res;
      if (R == 1) res = (sc * (logdist[0] + v[0]));
      else {
        double s = 0.0;
        for (int i = 0; i < R; i++) {
          double dist = sc * (minlogdist - logdist[i]);
          if (dist > -100.0) s += exp(dist - sc * v[i]);
        }
        res = (float) (sc * minlogdist - log(s));
      }
      score[(size_t) t * cb->K + k] = res;
    }
  }
  free(logdist);
}
