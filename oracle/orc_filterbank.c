/*
 * oracle/orc_filterbank.c -- TEST INFRASTRUCTURE (see orc.h).
 * Literal CPU restatement of the reference's uniform DFT filter banks:
 *   btk/modulated/modulated.h:79-163   (_RealBuffer ring)
 *   btk/modulated/modulated.cc:262-311 (OverSampledDFTFilterBank ctor: delays)
 *   btk/modulated/modulated.cc:400-516 (OverSampledDFTAnalysisBank)
 *   btk/modulated/modulated.cc:586-664 (OverSampledDFTSynthesisBank)
 *   btk/modulated/modulated.cc:72-97,121-257 (getWindow, NormalFFTAnalysisBank)
 * The FFT is GSL's radix-2 in the reference (a third-party dependency, GSL >= 1.10,
 * not vendored); it is mathematically the unnormalised DFT, restated here as a plain
 * iterative radix-2 in double.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* unnormalised DFT, sign=+1: e^{+2pi j kn/N} (gsl_fft_complex_radix2_backward),
   sign=-1: forward.  data interleaved re,im. N power of two. */
void orc_fft_radix2(double* data, int N, int sign)
{
  int j = 0;
  for (int i = 0; i < N - 1; i++) {
    if (i < j) {
      double tr = data[2*i], ti = data[2*i+1];
      data[2*i] = data[2*j]; data[2*i+1] = data[2*j+1];
      data[2*j] = tr; data[2*j+1] = ti;
    }
    int k = N >> 1;
    while (k <= j) { j -= k; k >>= 1; }
    j += k;
  }
  for (int len = 2; len <= N; len <<= 1) {
    int half = len >> 1;
    double theta = sign * 2.0 * M_PI / (double) len;
    for (int b = 0; b < half; b++) {
      double wr = cos(theta * b), wi = sin(theta * b);
      for (int i = b; i < N; i += len) {
        int jx = i + half;
        double xr = data[2*jx] * wr - data[2*jx+1] * wi;
        double xi = data[2*jx] * wi + data[2*jx+1] * wr;
        data[2*jx]   = data[2*i]   - xr;
        data[2*jx+1] = data[2*i+1] - xi;
        data[2*i]   += xr;
        data[2*i+1] += xi;
      }
    }
  }
}

/* ---- _RealBuffer (modulated.h:79-163) ---- */
typedef struct { int len, nsamp, zero; double* s; } ring_t;
static void ring_init(ring_t* r, int len, int nsamp)
{ r->len = len; r->nsamp = nsamp; r->zero = nsamp - 1; r->s = (double*) calloc((size_t) len * nsamp, sizeof(double)); }
static void ring_free(ring_t* r) { free(r->s); }
static double ring_sample(const ring_t* r, int timeX, int binX)
{ int idx = (r->zero + r->nsamp - timeX) % r->nsamp; return r->s[(size_t) idx * r->len + binX]; }
static double* ring_next(ring_t* r) { r->zero = (r->zero + 1) % r->nsamp; return r->s + (size_t) r->zero * r->len; }
static void ring_push(ring_t* r, const double* v, int reverse)
{
  double* b = ring_next(r);
  if (!v) { memset(b, 0, sizeof(double) * r->len); return; }
  if (reverse) for (int i = 0; i < r->len; i++) b[i] = v[r->len - i - 1];
  else memcpy(b, v, sizeof(double) * r->len);
}
static void ring_push_f(ring_t* r, const float* v)
{ double* b = ring_next(r); for (int i = 0; i < r->len; i++) b[i] = v[i]; }

/* modulated.cc:279-296 */
int orc_fb_processing_delay(int m, int r, int dctype, int synthesis)
{
  int R = 1 << r;
  switch (dctype) {
  case 1: return m * R - 1;
  case 2: return synthesis ? m * R / 2 : m * R - 1;
  default: return 2 * m - 1;
  }
}
int orc_fb_lookahead(int m, int r, int dctype, int synthesis)
{
  int R = 1 << r;
  if (dctype == 2 && !synthesis) return m * R / 2 - 1;
  return 0;
}

/* SampleFeature with padZeros=true yields ceil(nsamp/D) blocks (feature.cc:610-659);
   the bank then pads _processingDelay zero frames (modulated.cc:461-516).
   With laN>0 the first laN source blocks are pre-consumed (modulated.cc:467-474). */
int orc_analysis_num_frames(int nsamp, int M, int m, int r, int dctype)
{
  int D = M >> r;
  int nblk = (nsamp + D - 1) / D;
  int laN = orc_fb_lookahead(m, r, dctype, 0);
  int pd = orc_fb_processing_delay(m, r, dctype, 0);
  /* fewer source blocks than the look-ahead: the pre-consumption loop itself
     throws jiterator_error on every call (modulated.cc:467-474) -> no frames */
  if (nblk < laN) return 0;
  return nblk - laN + pd;
}

static void src_block(const float* x, int nsamp, int D, int blk, float* out)
{
  /* SampleFeature::next with blockLen=shiftLen=D, padZeros (feature.cc:627-653) */
  long cur = (long) blk * D;
  for (int i = 0; i < D; i++) out[i] = (cur + i < nsamp) ? x[cur + i] : 0.0f;
}

void orc_analysis_bank(const float* x, int nsamp, const double* h, int M, int m, int r,
                       int dctype, int gain, double* X)
{
  const int R = 1 << r, D = M / R;
  const int nblk = (nsamp + D - 1) / D;
  const int laN = orc_fb_lookahead(m, r, dctype, 0);
  const int T = orc_analysis_num_frames(nsamp, M, m, r, dctype);
  ring_t buffer, gsi;
  ring_init(&buffer, M, m * R);   /* _buffer(_M, m*_R)  modulated.cc:265 */
  ring_init(&gsi, D, R);          /* _gsi(_D, _R) */
  double* convert = (double*) calloc(M, sizeof(double));
  double* po = (double*) calloc(2 * M, sizeof(double));
  float* blk = (float*) calloc(D, sizeof(float));
  int srcX = 0;

#define UPDATE_BUF() do { \
    for (int sampX = 0; sampX < R; sampX++) \
      for (int dimX = 0; dimX < D; dimX++) \
        convert[dimX + sampX * D] = ring_sample(&gsi, R - sampX - 1, dimX); \
    ring_push(&buffer, convert, 1); } while (0)   /* modulated.cc:400-410 */

  /* look-ahead pre-consumption, modulated.cc:467-474 */
  for (int i = 0; i < laN && srcX < nblk; i++) {
    src_block(x, nsamp, D, srcX++, blk); ring_push_f(&gsi, blk); UPDATE_BUF();
  }
  for (int t = 0; t < T; t++) {
    if (srcX < nblk) { src_block(x, nsamp, D, srcX++, blk); ring_push_f(&gsi, blk); }
    else ring_push(&gsi, NULL, 0);            /* zero padding frames, modulated.cc:493-511 */
    UPDATE_BUF();
    /* polyphase, modulated.cc:419-434 */
    for (int k = 0; k < M; k++) {
      double sum = 0.0;
      for (int q = 0; q < m; q++) sum += h[k + M * q] * ring_sample(&buffer, R * q, k);
      po[2*k] = sum; po[2*k+1] = 0.0;
    }
    orc_fft_radix2(po, M, +1);                /* gsl_fft_complex_radix2_backward :439 */
    double* out = X + (size_t) t * 2 * M;
    for (int k = 0; k < M; k++) { out[2*k] = po[2*k]; out[2*k+1] = po[2*k+1]; }
    if (gain > 0)                             /* :444-448 */
      for (int k = 0; k < 2 * M; k++) out[k] = out[k] * (double) gain;
  }
#undef UPDATE_BUF
  free(convert); free(po); free(blk); ring_free(&buffer); ring_free(&gsi);
}

int orc_synthesis_bank(const double* Y, int T, const double* g, int M, int m, int r,
                       int dctype, int gain, float* out)
{
  const int R = 1 << r, D = M / R;
  const int pd = orc_fb_processing_delay(m, r, dctype, 1);
  ring_t buffer, gsi;
  ring_init(&buffer, M, m * R);
  ring_init(&gsi, M, R);          /* synthesis: _gsi(_M, _R)  modulated.cc:265 */
  double* convert = (double*) calloc(M, sizeof(double));
  double* pin = (double*) calloc(2 * M, sizeof(double));
  int nout = 0;

#define LOAD_FRAME(fx) do { \
    memcpy(pin, Y + (size_t)(fx) * 2 * M, sizeof(double) * 2 * M); \
    orc_fft_radix2(pin, M, -1);                 /* forward, modulated.cc:603 */ \
    for (int k = 0; k < M; k++) convert[k] = pin[2*k];   /* real part :606-607 */ \
    ring_push(&buffer, convert, 0); } while (0)

  /* priming, modulated.cc:631-634: frames 0..pd-1 (throws if the source is shorter) */
  if (T < pd) { free(convert); free(pin); ring_free(&buffer); ring_free(&gsi); return 0; }
  for (int i = 0; i < pd; i++) LOAD_FRAME(i);
  for (int t = 0; t + pd < T; t++) {
    LOAD_FRAME(t + pd);                         /* :639-642 */
    for (int k = 0; k < M; k++) {               /* :646-651 */
      double sum = 0.0;
      for (int q = 0; q < m; q++) sum += g[(M - k - 1) + M * q] * ring_sample(&buffer, R * q, k);
      convert[k] = sum;
    }
    ring_push(&gsi, convert, 0);
    float* o = out + (size_t) t * D;            /* :654-658, accumulation in the float vector */
    for (int d = 0; d < D; d++) o[d] = 0.0f;
    for (int sampX = 0; sampX < R; sampX++)
      for (int d = 0; d < D; d++)
        o[D - d - 1] = (float) ((double) o[D - d - 1] + ring_sample(&gsi, R - sampX - 1, d + sampX * D));
    if (gain > 0) for (int d = 0; d < D; d++) o[d] = o[d] * (float) gain;   /* :660-661 */
    nout++;
  }
#undef LOAD_FRAME
  free(convert); free(pin); ring_free(&buffer); ring_free(&gsi);
  return nout;
}

/* ---- PerfectReconstructionFFTAnalysisBank (modulated.cc:686-818) ----
 * 2M bands; prototype of length 2M*m; input blocks of D = M/R samples; T = nblk + (2m-1) frames (zero padding at the end).
 * X: [T][2M] interleaved complex double. */
int orc_pr_analysis_num_frames(int nsamp, int M, int m, int r)
{ int D = M >> r; int nblk = (nsamp + D - 1) / D; return nblk + (2 * m - 1); }
void orc_pr_analysis_bank(const float* x, int nsamp, const double* h, int M, int m, int r, double* X)
{
  const int R = 1 << r, D = M / R, M2 = 2 * M, R2 = 2 * R;
  const int nblk = (nsamp + D - 1) / D, T = nblk + (2 * m - 1);
  ring_t buffer, gsi;
  ring_init(&buffer, M2, m * (r + 2));          /* _buffer(_Mx2, m * (_r + 2))  modulated.cc:317 */
  ring_init(&gsi, D, R2);                       /* _gsi(_D, _Rx2) */
  double* convert = (double*) calloc(M2, sizeof(double));
  double* w = (double*) calloc(2 * M2, sizeof(double));
  float* blk = (float*) calloc(D, sizeof(float));
  { double vr = 1.0, vi = 0.0; const double cr = cos(-M_PI / (2.0 * M)), ci = sin(-M_PI / (2.0 * M));     /* w_k by repeated multiplication :693-699 */
    for (int k = 0; k < M2; k++) { w[2*k] = vr; w[2*k+1] = vi; const double nr = vr * cr - vi * ci, ni = vr * ci + vi * cr; vr = nr; vi = ni; } }
  for (int t = 0; t < T; t++) {
    if (t < nblk) { src_block(x, nsamp, D, t, blk); ring_push_f(&gsi, blk); }
    else ring_push(&gsi, NULL, 0);
    for (int sampX = 0; sampX < R2; sampX++)                                                              /* :722-728 */
      for (int dimX = 0; dimX < D; dimX++)
        convert[dimX + sampX * D] = ring_sample(&gsi, R2 - sampX - 1, dimX);
    ring_push(&buffer, convert, 1);
    double* out = X + (size_t) t * 2 * M2;
    for (int mm = 0; mm < M2; mm++) {                                                                     /* :737-750 */
      double sum = 0.0; int flip = 1;
      for (int k = 0; k < m; k++) { sum += flip * h[mm + M2 * k] * ring_sample(&buffer, (r + 2) * k, mm); flip *= -1; }
      out[2*mm] = w[2*mm] * sum; out[2*mm+1] = w[2*mm+1] * sum;
    }
    orc_fft_radix2(out, M2, +1);                                                                          /* gsl_fft_complex_radix2_inverse :759 */
    for (int i = 0; i < 2 * M2; i++) out[i] /= (double) M2;
  }
  free(convert); free(w); free(blk); ring_free(&buffer); ring_free(&gsi);
}

/* ---- PerfectReconstructionFFTSynthesisBank (modulated.cc:820-970) ----
 * Y: [T][2M] complex double in; out: [(T - (2m-1))][D] float; returns the number of output blocks */
int orc_pr_synthesis_bank(const double* Y, int T, const double* g, int M, int m, int r, float* out)
{
  const int R = 1 << r, D = M / R, M2 = 2 * M, R2 = 2 * R, pd = 2 * m - 1;
  ring_t buffer, gsi;
  ring_init(&buffer, M2, m * (r + 2));
  ring_init(&gsi, M2, R2);                      /* synthesis: _gsi(_Mx2, _Rx2) */
  double* convert = (double*) calloc(M2, sizeof(double));
  double* pin = (double*) calloc(2 * M2, sizeof(double));
  double* w = (double*) calloc(2 * M2, sizeof(double));
  { double vr = 1.0, vi = 0.0; const double cr = cos(M_PI / (2.0 * M)), ci = sin(M_PI / (2.0 * M));
    for (int k = 0; k < M2; k++) { w[2*k] = vr; w[2*k+1] = vi; const double nr = vr * cr - vi * ci, ni = vr * ci + vi * cr; vr = nr; vi = ni; } }
  int nout = 0, fed = 0;
#define FEED() do { memcpy(pin, Y + (size_t) fed * 2 * M2, sizeof(double) * 2 * M2); fed++; \
    orc_fft_radix2(pin, M2, -1);                                        /* forward :903 */ \
    for (int mm = 0; mm < M2; mm++) convert[mm] = pin[2*mm] * w[2*mm] - pin[2*mm+1] * w[2*mm+1];   /* Re(val * w) :907-910 */ \
    ring_push(&buffer, convert, 0); } while (0)
  if (T >= pd) {
    for (int i = 0; i < pd; i++) FEED();                                /* "prime" the buffer :921-924 */
    while (fed < T) {
      FEED();
      for (int mm = 0; mm < M2; mm++) {                                 /* :933-943 */
        double sum = 0.0; int flip = (m % 2 == 1) ? 1 : -1;
        for (int k = 0; k < m; k++) { sum += flip * g[mm + M2 * (m - k - 1)] * ring_sample(&buffer, (r + 2) * k, mm); flip *= -1; }
        convert[mm] = sum;
      }
      ring_push(&gsi, convert, 0);
      float* o = out + (size_t) nout * D;
      for (int d = 0; d < D; d++) o[d] = 0.0f;
      for (int sampX = 0; sampX < R2; sampX++)                          /* :947-950: float accumulation in the output vector */
        for (int d = 0; d < D; d++)
          o[D - d - 1] = (float) (o[D - d - 1] + ring_sample(&gsi, R2 - sampX - 1, d + sampX * D) / R);
      nout++;
    }
  }
#undef FEED
  free(convert); free(pin); free(w); ring_free(&buffer); ring_free(&gsi);
  return nout;
}

/* modulated.cc:72-97 */
void orc_get_window(int winType, int winLen, double* win)
{
  switch (winType) {
  case 0: for (int i = 0; i < winLen; i++) win[i] = 1.0; break;
  case 2: for (int i = 0; i < winLen; i++) win[i] = 0.5 * (1 - cos((2.0 * M_PI * i) / (double)(winLen - 1))); break;
  default: { double temp = 2. * M_PI / (double)(winLen - 1);
    for (int i = 0; i < winLen; i++) win[i] = 0.54 - 0.46 * cos(temp * i); } break;
  }
}

/* NormalFFTAnalysisBank: _processingDelay = _mx2 - 1 with m=1 -> 1 (modulated.cc:121-131) */
int orc_normal_fft_num_frames(int nsamp, int M, int r)
{
  int D = M >> r; int nblk = (nsamp + D - 1) / D; return nblk + 1;
}
void orc_normal_fft_bank(const float* x, int nsamp, int M, int r, int winType, double* X)
{
  const int R = 1 << r, D = M / R;
  const int nblk = (nsamp + D - 1) / D;
  const int T = nblk + 1;
  ring_t buffer, gsi;
  ring_init(&buffer, M, R); ring_init(&gsi, D, R);
  double* convert = (double*) calloc(M, sizeof(double));
  double* win = (double*) calloc(M, sizeof(double));
  float* blk = (float*) calloc(D, sizeof(float));
  orc_get_window(winType, M, win);
  for (int t = 0; t < T; t++) {
    if (t < nblk) { src_block(x, nsamp, D, t, blk); ring_push_f(&gsi, blk); }
    else ring_push(&gsi, NULL, 0);
    for (int sampX = 0; sampX < R; sampX++)
      for (int dimX = 0; dimX < D; dimX++)
        convert[dimX + sampX * D] = ring_sample(&gsi, R - sampX - 1, dimX);
    ring_push(&buffer, convert, 1);
    double* out = X + (size_t) t * 2 * M;
    for (int k = 0; k < M; k++) {        /* modulated.cc:233-244 */
      out[2*k] = win[k] * ring_sample(&buffer, 0, M - k - 1); out[2*k+1] = 0.0;
    }
    orc_fft_radix2(out, M, -1);          /* gsl_fft_complex_radix2_forward :250 */
  }
  free(convert); free(win); free(blk); ring_free(&buffer); ring_free(&gsi);
}
