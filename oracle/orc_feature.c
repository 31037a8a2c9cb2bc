/*
 * oracle/orc_feature.c -- TEST INFRASTRUCTURE (see orc.h).
 * CPU restatement of the MFCC feature operators, each in the precision the
 * reference uses (float/double mix):
 *   btk/feature/feature.cc:610-659    SampleFeature::next
 *   btk/feature/feature.cc:901-1020   BlockSizeConversionFeature
 *   btk/feature/feature.cc:1154-1170  PreemphasisFeature::next
 *   btk/feature/feature.cc:1206-1232  HammingFeature
 *   btk/feature/feature.cc:46-60,1266-1293 halfComplexUnpack, FFTFeature::next
 *   btk/feature/feature.cc:1329-1355  SpectralPowerFeature::next
 *   btk/feature/feature.cc:1716-1838  VTLNFeature::nextOrg / nextFF
 *   btk/feature/feature.cc:1942-2160  MelFeature::_SparseMatrix (melScaleOrg/FF, fmatrixBMulot)
 *   btk/feature/feature.cc:2398-2434  LogFeature::next
 *   btk/feature/feature.cc:2442-2490  CepstralFeature (+ btk/matrix/gslmatrix.cc:108-132)
 *   btk/feature/feature.cc:2573-2744  MeanSubtractionFeature (batch + run-on)
 *   btk/feature/feature.cc:2850-2927  AdjacentFeature
 *   btk/feature/feature.cc:2943-2957  LinearTransformFeature::next
 * gsl_blas_sgemv is restated as the GSL reference CBLAS loop (float accumulator,
 * ascending column index).  Compile with -ffp-contract=off.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_fft_radix2(double* data, int N, int sign);

int orc_sample_num_blocks(int nsamp, int blockLen, int shiftLen, int padZeros)
{
  /* feature.cc:619-653: stop when _cur >= N; a block with _cur+size >= N is padded or ends */
  int T = 0; long cur = 0;
  for (;;) {
    if (cur >= nsamp) break;
    if (cur + blockLen >= nsamp) { if (!padZeros) break; }
    T++; cur += shiftLen;
  }
  return T;
}

void orc_sample_blocks(const float* x, int nsamp, int blockLen, int shiftLen, int padZeros, float* out)
{
  int T = orc_sample_num_blocks(nsamp, blockLen, shiftLen, padZeros);
  for (int t = 0; t < T; t++) {
    long cur = (long) t * shiftLen;
    for (int i = 0; i < blockLen; i++) out[(size_t) t * blockLen + i] = (cur + i < nsamp) ? x[cur + i] : 0.0f;
  }
}

/* BlockSizeConversionFeature, emulated literally over a finite list of source blocks;
   a pull past the last block ends the stream (jiterator_error propagates). */
typedef struct { const float* in; int nIn, inLen, srcX; } bsc_src;
static const float* bsc_next(bsc_src* s) { s->srcX++; if (s->srcX >= s->nIn) return NULL; return s->in + (size_t) s->srcX * s->inLen; }

static int bsc_run(const float* in, int nIn, int inLen, int blockLen, int shiftLen, float* out)
{
  bsc_src s = { in, nIn, inLen, -1 };
  const int overlap = blockLen - shiftLen;
  float* vec = (float*) calloc(blockLen, sizeof(float));
  const float* feat = NULL; int curIn = 0, curOut = 0, frames = 0, first = 1;
  for (;;) {
    if (inLen > shiftLen) {                       /* _inputLonger, feature.cc:922-955 */
      if (first) {
        feat = bsc_next(&s); if (!feat) break;
        memcpy(vec, feat, sizeof(float) * blockLen); curIn += blockLen;
      } else {
        if (overlap > 0) memmove(vec, vec + shiftLen, sizeof(float) * overlap);
        if (curIn + shiftLen < inLen) {
          memcpy(vec + overlap, feat + curIn, sizeof(float) * shiftLen); curIn += shiftLen;
        } else {
          int remaining = inLen - curIn;
          if (remaining > 0) memcpy(vec + overlap, feat + curIn, sizeof(float) * remaining);
          curIn = 0;
          feat = bsc_next(&s); if (!feat) break;
          int fromNew = shiftLen - remaining;
          memcpy(vec + overlap + remaining, feat + curIn, sizeof(float) * fromNew); curIn += fromNew;
        }
      }
    } else {                                      /* _outputLonger, feature.cc:957-1007 */
      int ended = 0;
      if (!first) {
        if (overlap > 0) { memmove(vec, vec + shiftLen, sizeof(float) * overlap); curOut += overlap; }
        if (curIn > 0) {
          int remaining = inLen - curIn;
          if (remaining > 0) { memcpy(vec + curOut, feat + curIn, sizeof(float) * remaining); curOut += remaining; }
          curIn = 0;
        }
      }
      while (curOut + inLen <= blockLen) {
        feat = bsc_next(&s); if (!feat) { ended = 1; break; }
        memcpy(vec + curOut, feat, sizeof(float) * inLen); curOut += inLen;
      }
      if (ended) break;
      int remaining = blockLen - curOut;
      if (remaining > 0) {
        feat = bsc_next(&s); if (!feat) break;
        memcpy(vec + curOut, feat, sizeof(float) * remaining); curIn += remaining;
      }
      curOut = 0;
    }
    first = 0;
    if (out) memcpy(out + (size_t) frames * blockLen, vec, sizeof(float) * blockLen);
    frames++;
  }
  free(vec);
  return frames;
}
int orc_blockconv_num_frames(int nIn, int inLen, int blockLen, int shiftLen)
{
  float* z = (float*) calloc((size_t) (nIn > 0 ? nIn : 1) * inLen, sizeof(float));
  int n = bsc_run(z, nIn, inLen, blockLen, shiftLen, NULL); free(z); return n;
}
void orc_blockconv(const float* in, int nIn, int inLen, int blockLen, int shiftLen, float* out)
{ bsc_run(in, nIn, inLen, blockLen, shiftLen, out); }

void orc_preemphasis(const float* in, int T, int L, double mu, float* out)
{
  float prior = 0.0f;                                /* float _prior, feature.h:475 */
  for (int t = 0; t < T; t++)
    for (int i = 0; i < L; i++) {
      float b = in[(size_t) t * L + i];
      out[(size_t) t * L + i] = (float) ((double) b - mu * (double) prior);
      prior = b;
    }
}

void orc_hamming(const float* in, int T, int L, float* out)
{
  double* w = (double*) malloc(sizeof(double) * L);
  double temp = 2. * M_PI / (double) (L - 1);
  for (int i = 0; i < L; i++) w[i] = 0.54 - 0.46 * cos(temp * i);
  for (int t = 0; t < T; t++)
    for (int i = 0; i < L; i++) out[(size_t) t * L + i] = (float) (w[i] * (double) in[(size_t) t * L + i]);
  free(w);
}

void orc_fft_feature(const float* in, int T, int L, int fftLen, double* out)
{
  /* real FFT (forward sign) then unpack to full complex with conjugate mirror */
  double* buf = (double*) malloc(sizeof(double) * 2 * fftLen);
  for (int t = 0; t < T; t++) {
    for (int i = 0; i < fftLen; i++) { buf[2*i] = (i < L) ? (double) in[(size_t) t * L + i] : 0.0; buf[2*i+1] = 0.0; }
    orc_fft_radix2(buf, fftLen, -1);
    double* o = out + (size_t) t * 2 * fftLen;
    int len2 = (fftLen + 1) / 2;
    o[0] = buf[0]; o[1] = 0.0;
    if ((fftLen & 1) == 0) { o[2*len2] = buf[2*len2]; o[2*len2+1] = 0.0; }
    for (int m = 1; m < len2; m++) {
      /* half-complex: src[m]=Re X[m], src[len-m]=Im X[m]; tgt[m]=(re,im), tgt[len-m]=(re,-im) */
      o[2*m] = buf[2*m]; o[2*m+1] = buf[2*m+1];
      o[2*(fftLen-m)] = buf[2*m]; o[2*(fftLen-m)+1] = -buf[2*m+1];
    }
  }
  free(buf);
}

void orc_spectral_power(const double* fft, int T, int fftLen, int powN, double* out)
{
  for (int t = 0; t < T; t++)
    for (int i = 0; i < powN; i++) {
      double re = fft[((size_t) t * fftLen + i) * 2], im = fft[((size_t) t * fftLen + i) * 2 + 1];
      out[(size_t) t * powN + i] = re * re + im * im;     /* gsl_complex_abs2 */
    }
}

void orc_vtln(const double* pw, int T, int N, double ratio, double edge, int version, double* out)
{
  if (version == 1) {                                  /* nextOrg feature.cc:1716-1766 */
    double yedge = (edge < ratio) ? (edge / ratio) : 1.0;
    double b = (yedge < 1.0) ? (1.0 - edge) / (1.0 - yedge) : 0;
    for (int t = 0; t < T; t++) {
      const double* p = pw + (size_t) t * N; double* o = out + (size_t) t * N;
      for (int cx = 0; cx < N; cx++) {
        double Y0 = (double) cx / (double) N, Y1 = (double) (cx + 1) / (double) N;
        double X0 = ((Y0 < yedge) ? (ratio * Y0) : (b * Y0 + 1.0 - b)) * N;
        double X1 = ((Y1 < yedge) ? (ratio * Y1) : (b * Y1 + 1.0 - b)) * N;
        int L1 = (int) X1; double alpha1 = X1 - L1;
        int L0 = (int) X0; double alpha0 = (int) X0 + 1 - X0;
        double z = 0.0;
        if (L0 >= N) L0 = N - 1;
        if (L1 > N) L1 = N;
        if (L0 == L1) z += (X1 - X0) * p[L0];
        else {
          z += alpha0 * p[L0];
          for (int i = L0 + 1; i < L1; i++) z += p[i];
          if (L1 < N) z += alpha1 * p[L1];
        }
        o[cx] = z;
      }
    }
  } else {                                             /* nextFF feature.cc:1769-1838 */
    double* aux = (double*) malloc(sizeof(double) * N);
    for (int t = 0; t < T; t++) {
      const double* p = pw + (size_t) t * N; double* o = out + (size_t) t * N;
      float b = N * edge; float slope1 = ratio, slope2 = ratio;
      if (slope1 < 1.0) slope2 = (N - slope1 * b) / (N - b);
      for (int i = 0; i < N; i++) { o[i] = 0.0; aux[i] = 0.0; }
      for (int sIdx = 0; sIdx < N; sIdx++) {
        float s1 = sIdx - 0.5, s2 = sIdx + 0.5; float v = p[sIdx];
        float d1 = s1 * slope1; if (s1 > b) d1 = b * slope1 + (s1 - b) * slope2;
        float d2 = s2 * slope1; if (s2 > b) d2 = b * slope1 + (s2 - b) * slope2;
        int i1 = (int) floor(d1), i2 = (int) ceil(d2);
        if (i1 <= N - 1) {
          double alpha = 1.0, alpha1 = (1.0 - (d1 - i1)) * alpha, alpha2 = (i2 - d2) * alpha;
          for (int j = i1; j <= i2; j++) {
            int k = j; if (k < 0) k = 0; if (k >= N) break;
            double a = alpha; if (j == i1) a = alpha1; if (j == i2) a = alpha2;
            o[k] = o[k] + a * v; aux[k] = aux[k] + a;
          }
        }
      }
      for (int i = 0; i < N; i++) { double norm = aux[i]; if (norm > 1E-20) o[i] = o[i] / norm; }
    }
    free(aux);
  }
}

static float mel_of(float hz) { if (hz >= 0) return (float) (2595.0 * log10(1.0 + (double) hz / 700.0)); else return 0.0; }
static float hertz_of(float m) { double d = m / 2595.0; return (float) (700.0 * (pow(10.0, d) - 1.0)); }

orc_melbank* orc_melbank_create(int powN, float rate, float low, float up, int filterN, int version)
{
  /* melScaleOrg / melScaleFF, feature.cc:1954-2090.  All edge maths in float. */
  if (up <= 0) up = rate / 2.0;                       /* MelFeature ctor :2262 */
  float df = rate / (4.0 * (powN / 2));
  float mlow = mel_of(low), mup = mel_of(up);
  float dm = (mup - mlow) / (filterN + 1);
  if (low < 0.0 || 2.0 * up > rate || low > up) return NULL;
  orc_melbank* mb = (orc_melbank*) calloc(1, sizeof(orc_melbank));
  mb->filterN = filterN;
  mb->offset = (int*) calloc(filterN, sizeof(int)); mb->coefN = (int*) calloc(filterN, sizeof(int));
  mb->data = (float**) calloc(filterN, sizeof(float*));
  for (int x = 0; x < filterN; x++) {
    float left = hertz_of(x * dm + mlow);
    float center = hertz_of((x + 1.0) * dm + mlow);
    float right = hertz_of((x + 2.0) * dm + mlow);
    float height = 2.0 / (right - left);
    float slope1 = height / (center - left);
    float slope2 = height / (center - right);
    int start = (int) ceil(left / df);
    int end = (int) floor(right / df);
    mb->offset[x] = start; mb->coefN[x] = end - start + 1; mb->n = end;
    mb->data[x] = (float*) calloc(mb->coefN[x] > 0 ? mb->coefN[x] : 1, sizeof(float));
    float freq = start * df;
    for (int i = 0; i < mb->coefN[x]; i++) {
      if (version == 1) freq += df;                   /* Org: add before use (:2005) */
      if (freq <= center) mb->data[x][i] = slope1 * (freq - left);
      else mb->data[x][i] = slope2 * (freq - right);
      if (version != 1) freq += df;                   /* FF fix (:2080) */
    }
  }
  return mb;
}
void orc_melbank_free(orc_melbank* mb)
{
  if (!mb) return;
  for (int i = 0; i < mb->filterN; i++) free(mb->data[i]);
  free(mb->data); free(mb->offset); free(mb->coefN); free(mb);
}

void orc_mel(const orc_melbank* mb, const double* pw, int T, int powN, int version, double* out)
{
  (void) version;   /* both fmatrixBMulot variants group the sum identically */
  for (int t = 0; t < T; t++) {
    const double* A = pw + (size_t) t * powN;
    for (int j = 0; j < mb->filterN; j++) {
      double sum = 0.0;
      const double* aP = A + mb->offset[j]; const float* bP = mb->data[j];
      int n = mb->coefN[j], i = 0;
      for (; i + 4 <= n; i += 4)
        sum += aP[i]*bP[i] + aP[i+1]*bP[i+1] + aP[i+2]*bP[i+2] + aP[i+3]*bP[i+3];
      for (; i < n; i++) sum += aP[i] * bP[i];
      out[(size_t) t * mb->filterN + j] = sum;
    }
  }
}

void orc_log(const double* mel, int T, int N, double m, double a, int sphinxFlooring, float* out)
{
  for (size_t i = 0; i < (size_t) T * N; i++) {
    double val = mel[i];
    if (sphinxFlooring) { if (val < 1.0E-05) val = 1.0E-05; }
    else { val += a; if (val <= 0.0) val = 1.0; }
    out[i] = (float) (m * log10(val));
  }
}

void orc_cosine_matrix(int ncep, int nmel, int type, float* Cm)
{
  if (type == 0) {                                    /* gslmatrix.cc:115-123 */
    for (int k = 0; k < ncep; k++) {
      double fac = k * M_PI / (double) (nmel - 1);
      float* p = Cm + (size_t) k * nmel;
      *p++ = 1.0;
      for (int l = 1; l < nmel - 1; l++) *p++ = 2.0 * cos(fac * l);
      *p = cos(k * M_PI);
    }
  } else if (type == 1) {                             /* :124-129 */
    for (int k = 0; k < ncep; k++) {
      double fac = k * M_PI / (double) nmel;
      for (int l = 0; l < nmel; l++) Cm[(size_t) k * nmel + l] = cos(fac * (l + 0.5));
    }
  } else {                                            /* _sphinxLegacy feature.cc:2466-2477 */
    for (int c = 0; c < ncep; c++) {
      double deltaF = M_PI * (float) c / nmel;
      for (int f = 0; f < nmel; f++) {
        double frequency = deltaF * (f + 0.5);
        double cv = cos(frequency) / nmel;
        if (f == 0) cv *= 0.5;
        Cm[(size_t) c * nmel + f] = cv;
      }
    }
  }
}

void orc_sgemv_rows(const float* A, int rows, int cols, const float* X, int T, float* Y)
{
  for (int t = 0; t < T; t++)
    for (int i = 0; i < rows; i++) {
      float temp = 0.0f;
      for (int j = 0; j < cols; j++) temp += X[(size_t) t * cols + j] * A[(size_t) i * cols + j];
      Y[(size_t) t * rows + i] = 0.0f + 1.0f * temp;
    }
}

void orc_cmn_batch(const float* in, int T, int N, double devNormFactor, float* out, float* mean, float* var)
{ orc_cmn_batch_w(in, T, N, devNormFactor, NULL, out, mean, var); }
void orc_cmn_batch_w(const float* in, int T, int N, double devNormFactor, const float* weights, float* out, float* mean, float* var)
{
  /* _calcMeanVariance, feature.cc:2633-2707: float accumulators, double total weight; weights: element 0 of the weight stream's frames (:2640-2644), NULL = 1 */
  float* mu = (float*) calloc(N, sizeof(float)); float* vr = (float*) calloc(N, sizeof(float));
  double ttl = 0.0;
  for (int t = 0; t < T; t++) { float wgt = weights ? weights[t] : 1.0f; for (int i = 0; i < N; i++) mu[i] = mu[i] + wgt * in[(size_t) t * N + i]; ttl += wgt; }
  for (int i = 0; i < N; i++) mu[i] = mu[i] / ttl;
  ttl = 0.0;
  for (int t = 0; t < T; t++) { float wgt = weights ? weights[t] : 1.0f; for (int i = 0; i < N; i++) { float f = in[(size_t) t * N + i]; vr[i] = vr[i] + wgt * f * f; } ttl += wgt; }
  for (int i = 0; i < N; i++) { float m = mu[i]; vr[i] = (vr[i] / ttl) - (m * m); }
  for (int t = 0; t < T; t++)
    for (int i = 0; i < N; i++) {
      float v = in[(size_t) t * N + i] - mu[i];
      if (devNormFactor > 0.0) { float va = vr[i]; if (va < 0.0001f) va = 0.0001f; v = v / (devNormFactor * sqrtf(va)); /* C++ sqrt(float) overload */ }
      out[(size_t) t * N + i] = v;
    }
  if (mean) memcpy(mean, mu, sizeof(float) * N);
  if (var) memcpy(var, vr, sizeof(float) * N);
  free(mu); free(vr);
}

void orc_cmn_runon(const float* in, int T, int N, double devNormFactor, float* out)
{ orc_cmn_runon_w(in, T, N, devNormFactor, NULL, out); }
void orc_cmn_runon_w(const float* in, int T, int N, double devNormFactor, const float* weights, float* out)
{
  /* _nextRunon, feature.cc:2577-2618: frames whose weight is not positive are normalised but do not update the statistics */
  float* mu = (float*) calloc(N, sizeof(float)); float* vr = (float*) malloc(sizeof(float) * N);
  for (int i = 0; i < N; i++) vr[i] = 1.0f;
  unsigned framesN = 0;
  for (int t = 0; t < T; t++) {
    const float* s = in + (size_t) t * N;
    if (!weights || weights[t] > 0.0) {
    float wgt = (framesN < 500) ? 0.98f : 0.995f;
    for (int i = 0; i < N; i++) { float comp = wgt * mu[i] + (1.0 - wgt) * s[i]; mu[i] = comp; }
    if (devNormFactor > 0.0)
      for (int i = 0; i < N; i++) { float diff = s[i] - mu[i]; float comp = wgt * vr[i] + (1.0 - wgt) * (diff * diff); vr[i] = comp; }
    framesN++;
    }
    for (int i = 0; i < N; i++) {
      float v = s[i] - mu[i];
      if (devNormFactor > 0.0) { float va = vr[i]; if (va < 0.0001f) va = 0.0001f; v = v / (devNormFactor * sqrtf(va)); /* C++ sqrt(float) overload */ }
      out[(size_t) t * N + i] = v;
    }
  }
  free(mu); free(vr);
}

int orc_adjacent(const float* in, int T, int N, int delta, float* out)
{
  /* closed form of the buffer shuffling in feature.cc:2850-2904: slot s of output frame t
     holds input frame clamp(t+s-delta, 0, T-1); the initial fill needs frames 0..delta-1,
     so a stream shorter than that ends before the first output frame. */
  if (T < delta || T < 1) return 0;
  for (int t = 0; t < T; t++)
    for (int s = 0; s <= 2 * delta; s++) {
      int src = t + s - delta; if (src < 0) src = 0; if (src > T - 1) src = T - 1;
      memcpy(out + ((size_t) t * (2 * delta + 1) + s) * N, in + (size_t) src * N, sizeof(float) * N);
    }
  return T;
}

void orc_mfcc_default_cfg(orc_mfcc_cfg* c)
{
  /* SWIG ctor defaults, btk/feature/feature.i:526-528,738,822,873,1084-1087,1118-1121,
     1177-1179,1240,1445-1447,1507-1509 and SURVEY.md Appendix C.3 */
  memset(c, 0, sizeof(*c));
  c->blockLen = 320; c->shiftLen = 160; c->padZeros = 0; c->mu = 0.95; c->fftLen = 512; c->powN = 257;
  c->vtlnRatio = 1.0; c->vtlnEdge = 1.0; c->vtlnVersion = 1;
  c->rate = 16000.0f; c->low = 0.0f; c->up = 0.0f; c->filterN = 30; c->melVersion = 1;
  c->logM = 1.0; c->logA = 1.0; c->ncep = 13; c->dctType = 1; c->devNormFactor = 0.0; c->delta = 7;
  c->outDim = 39; c->lda = NULL;
}

int orc_mfcc_num_frames(const orc_mfcc_cfg* c, int nsamp)
{
  int T = orc_sample_num_blocks(nsamp, c->blockLen, c->shiftLen, c->padZeros);
  if (c->delta > 0 && T < c->delta) return 0;
  return T;
}

int orc_mfcc_from_blocks(const orc_mfcc_cfg* c, const float* blocks, int T, int stage, float* out)
{
  const int L = c->blockLen, F = c->fftLen, P = c->powN, NM = c->filterN, NC = c->ncep;
  if (T <= 0) return 0;
  float* pre = (float*) malloc(sizeof(float) * (size_t) T * L);
  float* ham = (float*) malloc(sizeof(float) * (size_t) T * L);
  double* fft = (double*) malloc(sizeof(double) * 2 * (size_t) T * F);
  double* pw = (double*) malloc(sizeof(double) * (size_t) T * P);
  double* vt = (double*) malloc(sizeof(double) * (size_t) T * P);
  double* mel = (double*) malloc(sizeof(double) * (size_t) T * NM);
  float* lg = (float*) malloc(sizeof(float) * (size_t) T * NM);
  float* cep = (float*) malloc(sizeof(float) * (size_t) T * NC);
  float* cmn = (float*) malloc(sizeof(float) * (size_t) T * NC);
  float* Cm = (float*) malloc(sizeof(float) * (size_t) NC * NM);
  orc_melbank* mb = orc_melbank_create(P, c->rate, c->low, c->up, NM, c->melVersion);
  int ret = T;
  orc_preemphasis(blocks, T, L, c->mu, pre);
  orc_hamming(pre, T, L, ham);
  orc_fft_feature(ham, T, L, F, fft);
  orc_spectral_power(fft, T, F, P, pw);
  orc_vtln(pw, T, P, c->vtlnRatio, c->vtlnEdge, c->vtlnVersion, vt);
  orc_mel(mb, vt, T, P, c->melVersion, mel);
  orc_log(mel, T, NM, c->logM, c->logA, c->sphinxFlooring, lg);
  orc_cosine_matrix(NC, NM, c->dctType, Cm);
  orc_sgemv_rows(Cm, NC, NM, lg, T, cep);
  orc_cmn_batch(cep, T, NC, c->devNormFactor, cmn, NULL, NULL);
  if (stage == 4) { for (size_t i = 0; i < (size_t) T * P; i++) out[i] = (float) pw[i]; }
  else if (stage == 3) memcpy(out, lg, sizeof(float) * (size_t) T * NM);
  else if (stage == 1) memcpy(out, cep, sizeof(float) * (size_t) T * NC);
  else if (stage == 2) memcpy(out, cmn, sizeof(float) * (size_t) T * NC);
  else {
    int W = (2 * c->delta + 1) * NC;
    float* adj = (float*) malloc(sizeof(float) * (size_t) T * W);
    int Ta = orc_adjacent(cmn, T, NC, c->delta, adj);
    if (c->lda) orc_sgemv_rows(c->lda, c->outDim, W, adj, Ta, out);
    else memcpy(out, adj, sizeof(float) * (size_t) Ta * W);
    ret = Ta; free(adj);
  }
  orc_melbank_free(mb);
  free(pre); free(ham); free(fft); free(pw); free(vt); free(mel); free(lg); free(cep); free(cmn); free(Cm);
  return ret;
}

int orc_mfcc_chain(const orc_mfcc_cfg* c, const float* x, int nsamp, int stage, float* out)
{
  int T = orc_sample_num_blocks(nsamp, c->blockLen, c->shiftLen, c->padZeros);
  if (T <= 0) return 0;
  float* blk = (float*) malloc(sizeof(float) * (size_t) T * c->blockLen);
  orc_sample_blocks(x, nsamp, c->blockLen, c->shiftLen, c->padZeros, blk);
  int r = orc_mfcc_from_blocks(c, blk, T, stage, out);
  free(blk);
  return r;
}
