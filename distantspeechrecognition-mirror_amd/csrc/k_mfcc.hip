// csrc/k_mfcc.hip -- the MFCC operator chain as three device kernels.
//
//   k_mfcc_frames : SampleFeature framing (feature.cc:610-659) -> PreemphasisFeature (:1154-1170)
//                   -> HammingFeature (:1206-1232) -> FFTFeature (:1266-1293) -> SpectralPowerFeature
//                   (:1329-1355) -> VTLNFeature (:1716-1838) -> MelFeature (:2098-2160) -> LogFeature
//                   (:2398-2434) -> CepstralFeature (:2479-2490).  One wavefront per frame, the
//                   length-fftLen real FFT as a length-fftLen/2 complex Stockham FFT in LDS.
//   k_cmn         : MeanSubtractionFeature batch / run-on statistics (:2586-2744)
//   k_splice_lda  : AdjacentFeature (:2850-2904) + LinearTransformFeature (:2943-2957)
//
// Precision follows the reference operator by operator (fp32 samples, fp64 FFT/power/VTLN/mel,
// fp32 log-mel, cepstra and transforms with sequential fp32 accumulation as gsl_blas_sgemv's
// reference loop), so the chain agrees with the CPU path to rounding of libm/FFT ordering only.
// Tables (Hamming, VTLN interval weights, mel triangles, DCT) are built on the host with the
// reference's own formulas, including its quirks (mel v1 evaluates the triangle one bin late).
#include "common.h"
#include "ops.h"
#include <cmath>

namespace dsr {

struct SparseRows {            // out[k] = (sum_i coef[off+i] * in[start+i]) [/ div]
  std::vector<int> start, count, off; std::vector<double> coef, div;
};

struct MfccPlan {
  dsr_mfcc_cfg c;
  int melN = 0;                // required input length of the mel bank (_n)
  DevBuf<double> d_ham;        // [blockLen]
  DevBuf<double2> d_tw;        // [fftLen] e^{+2 pi j k / fftLen}
  DevBuf<int> d_vStart, d_vCount, d_vOff; DevBuf<double> d_vCoef, d_vDiv;
  DevBuf<int> d_mStart, d_mCount, d_mOff; DevBuf<float> d_mCoef;
  DevBuf<float> d_dct;         // [ncep][filterN]
  int melCoefN = 0;            // entries of d_mCoef
  DevBuf<float> d_lda;         // [outDim][(2delta+1)*ncep]
  DevBuf<float> w_cep, w_cmn;  // workspaces [U][Tmax][ncep]
  DevBuf<float> w_pow, w_logmel;
  int vtlnRoundFloat = 0;
};

// (a wavefront's LDS instructions execute in issue order: when the buffers belong to one wavefront a compiler fence is all a stage boundary needs)
template <bool WAVE> __device__ __forceinline__ void stage_sync()
{
  if (WAVE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
  else __syncthreads();
}

template <int N, bool WAVE = false>
__device__ __forceinline__ double2* fft_lds_d(double2* x, double2* y, const double2* tw, int twStep, int sign, int lane, int nl)
{
  int n = N, s = 1; const double sj = (double) sign;
  while (n >= 4) {
    const int m4 = n >> 2; const int twn = (N / n) * twStep;
    for (int j = lane; j < N / 4; j += nl) {
      const int p = j / s, q = j - p * s;
      const double2 a = x[q + s * p], b = x[q + s * (p + m4)], c = x[q + s * (p + 2 * m4)], d = x[q + s * (p + 3 * m4)];
      const double2 apc = make_double2(a.x + c.x, a.y + c.y), amc = make_double2(a.x - c.x, a.y - c.y);
      const double2 bpd = make_double2(b.x + d.x, b.y + d.y), bmd = make_double2(b.x - d.x, b.y - d.y);
      const double2 jb = make_double2(-sj * bmd.y, sj * bmd.x);
      double2 w1 = tw[p * twn], w2 = tw[2 * p * twn], w3 = tw[3 * p * twn];
      w1.y *= sj; w2.y *= sj; w3.y *= sj;
      const double2 t1 = make_double2(amc.x + jb.x, amc.y + jb.y), t2 = make_double2(apc.x - bpd.x, apc.y - bpd.y);
      const double2 t3 = make_double2(amc.x - jb.x, amc.y - jb.y);
      y[q + s * (4 * p + 0)] = make_double2(apc.x + bpd.x, apc.y + bpd.y);
      y[q + s * (4 * p + 1)] = make_double2(t1.x * w1.x - t1.y * w1.y, t1.x * w1.y + t1.y * w1.x);
      y[q + s * (4 * p + 2)] = make_double2(t2.x * w2.x - t2.y * w2.y, t2.x * w2.y + t2.y * w2.x);
      y[q + s * (4 * p + 3)] = make_double2(t3.x * w3.x - t3.y * w3.y, t3.x * w3.y + t3.y * w3.x);
    }
    stage_sync<WAVE>();
    double2* t = x; x = y; y = t; n >>= 2; s <<= 2;
  }
  if (n == 2) {
    for (int q = lane; q < N / 2; q += nl) {
      const double2 a = x[q], b = x[q + s];
      y[q] = make_double2(a.x + b.x, a.y + b.y); y[q + s] = make_double2(a.x - b.x, a.y - b.y);
    }
    stage_sync<WAVE>();
    double2* t = x; x = y; y = t;
  }
  return x;
}

struct MfccDev {
  int blockLen, shiftLen, padZeros, fftLen, powN, filterN, ncep, sphinx, vtlnOn, preOn, vtlnRoundFloat;
  double mu, logM, logA;
  const double* ham; const double2* tw;
  const int *vStart, *vCount, *vOff; const double *vCoef, *vDiv;
  const int *mStart, *mCount, *mOff; const float* mCoef;
  const float* dct;
};

// One wavefront (64 lanes) per frame, FPB frames per workgroup.
// LDS per frame: 2 x (FFTN/2) double2 ping-pong.
template <int FFTN>
__global__ __launch_bounds__(256) void k_mfcc_frames(MfccDev P, const float* __restrict__ y, const int* __restrict__ nsampArr,
                                                     long sampStride, int Tmax, float* __restrict__ cep,
                                                     float* __restrict__ powOut, float* __restrict__ logmelOut)
{
  constexpr int N = FFTN / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, FPB = blockDim.x >> 6;
  double2* bufA = reinterpret_cast<double2*>(smem) + (size_t) wave * 2 * N;
  double2* bufB = bufA + N;
  const int u = blockIdx.y;
  const int t = blockIdx.x * FPB + wave;
  const int nsamp = nsampArr[u];
  // SampleFeature frame count (feature.cc:619-653)
  int Tu;
  if (P.padZeros) Tu = (nsamp + P.shiftLen - 1) / P.shiftLen;
  else { long a = (long) nsamp - P.blockLen; Tu = (a > 0) ? (int) ((a + P.shiftLen - 1) / P.shiftLen) : 0; }
  const bool live = (t < Tu) && (t < Tmax);
  const float* ys = y + (long) u * sampStride;
  const long cur = (long) t * P.shiftLen;

  // 1. framing + pre-emphasis + Hamming -> packed complex z[n] = (s[2n], s[2n+1])
  double* zr = reinterpret_cast<double*>(bufA);
  for (int i = lane; i < FFTN; i += 64) {
    double v = 0.0;
    if (live && i < P.blockLen) {
      const long n = cur + i;
      const float b = (n < nsamp) ? ys[n] : 0.0f;
      float pre = b;
      if (P.preOn) {
        float prior;
        if (i > 0) { const long n1 = n - 1; prior = (n1 < nsamp) ? ys[n1] : 0.0f; }
        else if (t == 0) prior = 0.0f;
        else { const long n1 = (long) (t - 1) * P.shiftLen + P.blockLen - 1; prior = (n1 < nsamp) ? ys[n1] : 0.0f; }
        pre = (float) __dsub_rn((double) b, __dmul_rn(P.mu, (double) prior));
      }
      const float hm = (float) __dmul_rn(P.ham[i], (double) pre);
      v = (double) hm;
    }
    zr[i] = v;
  }
  __syncthreads();
  // 2. complex FFT of length N (forward sign), then even/odd split -> power spectrum
  double2* Z = fft_lds_d<N>(bufA, bufB, P.tw, 2, -1, lane, 64);
  double* pw = reinterpret_cast<double*>(Z == bufA ? bufB : bufA);     // free buffer: powN doubles fit (powN <= FFTN)
  for (int f = lane; f < P.powN; f += 64) {
    // bins above N mirror: |X[FFTN-f]| = |X[f]| (halfComplexUnpack writes the conjugate, feature.cc:55-58)
    const int ff = (f <= N) ? f : (FFTN - f);
    const double2 zf = Z[ff & (N - 1)]; double2 zc = Z[(N - ff) & (N - 1)]; zc.y = -zc.y;
    const double2 E = make_double2(0.5 * (zf.x + zc.x), 0.5 * (zf.y + zc.y));
    const double2 dd = make_double2(zf.x - zc.x, zf.y - zc.y);
    const double2 O = make_double2(0.5 * dd.y, -0.5 * dd.x);
    double2 w = P.tw[ff]; w.y = -w.y;
    const double re = E.x + (w.x * O.x - w.y * O.y), im = E.y + (w.x * O.y + w.y * O.x);
    pw[f] = __dadd_rn(__dmul_rn(re, re), __dmul_rn(im, im));
  }
  __syncthreads();
  if (powOut && live) for (int f = lane; f < P.powN; f += 64) powOut[((long) u * Tmax + t) * P.powN + f] = (float) pw[f];
  // 3. VTLN (sparse interval weights) -> vt (other buffer)
  double* vt = reinterpret_cast<double*>(Z);       // Z no longer needed
  if (P.vtlnOn) {
    for (int k = lane; k < P.powN; k += 64) {
      const int s0 = P.vStart[k], n = P.vCount[k], o = P.vOff[k];
      double z = 0.0;
      for (int i = 0; i < n; i++) {
        double in = pw[s0 + i]; if (P.vtlnRoundFloat) in = (double) (float) in;
        z = __dadd_rn(z, __dmul_rn(P.vCoef[o + i], in));
      }
      const double dv = P.vDiv[k]; if (dv != 0.0) z = z / dv;
      vt[k] = z;
    }
  } else {
    for (int k = lane; k < P.powN; k += 64) vt[k] = pw[k];
  }
  __syncthreads();
  // 4. mel (double accumulate, 4-grouped as fmatrixBMulot) + log10 -> float in LDS (reuse pw as float area)
  float* lg = reinterpret_cast<float*>(pw);
  for (int j = lane; j < P.filterN; j += 64) {
    const int s0 = P.mStart[j], n = P.mCount[j], o = P.mOff[j];
    double sum = 0.0; int i = 0;
    for (; i + 4 <= n; i += 4) {
      double g4 = __dmul_rn(vt[s0 + i], (double) P.mCoef[o + i]);
      g4 = __dadd_rn(g4, __dmul_rn(vt[s0 + i + 1], (double) P.mCoef[o + i + 1]));
      g4 = __dadd_rn(g4, __dmul_rn(vt[s0 + i + 2], (double) P.mCoef[o + i + 2]));
      g4 = __dadd_rn(g4, __dmul_rn(vt[s0 + i + 3], (double) P.mCoef[o + i + 3]));
      sum = __dadd_rn(sum, g4);
    }
    for (; i < n; i++) sum = __dadd_rn(sum, __dmul_rn(vt[s0 + i], (double) P.mCoef[o + i]));
    double val = sum;
    if (P.sphinx) { if (val < 1.0E-05) val = 1.0E-05; }
    else { val = __dadd_rn(val, P.logA); if (val <= 0.0) val = 1.0; }
    lg[j] = (float) __dmul_rn(P.logM, log10(val));
  }
  __syncthreads();
  if (logmelOut && live) for (int j = lane; j < P.filterN; j += 64) logmelOut[((long) u * Tmax + t) * P.filterN + j] = lg[j];
  // 5. DCT, sequential fp32 accumulation (gsl_blas_sgemv reference loop)
  if (t < Tmax)
    for (int k = lane; k < P.ncep; k += 64) {
      float temp = 0.0f;
      if (live) for (int j = 0; j < P.filterN; j++) temp = __fadd_rn(temp, __fmul_rn(lg[j], P.dct[k * P.filterN + j]));
      cep[((long) u * Tmax + t) * P.ncep + k] = live ? __fadd_rn(0.0f, temp) : 0.0f;
    }
}

// The same chain with the tables in LDS and wave-private frames: a workgroup stages twiddles, window, mel triangles and the DCT matrix
// once and each of its four wavefronts then walks FW consecutive frames on its own pair of buffers -- no workgroup barrier after the
// staging, no table read from memory inside the frame loop (the first kernel spends most of a frame waiting on those: every mel group and
// every DCT term is a dependent global load).  Same arithmetic, same order: the two kernels return the same bits.  VTLN tables stay in memory.
template <int FFTN, int FW>
__global__ __launch_bounds__(256) void k_mfcc_frames_w(MfccDev P, int melCoefN, const float* __restrict__ y, const int* __restrict__ nsampArr,
                                                       long sampStride, int Tmax, float* __restrict__ cep,
                                                       float* __restrict__ powOut, float* __restrict__ logmelOut)
{
  constexpr int N = FFTN / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double2* twL = reinterpret_cast<double2*>(smem);                                         // [FFTN]
  double2* bufs = twL + FFTN;                                                              // [4][2][N]
  double* hamL = reinterpret_cast<double*>(bufs + 4 * 2 * N);                              // [blockLen]
  float* mCoefL = reinterpret_cast<float*>(hamL + P.blockLen);                             // [melCoefN]
  float* dctL = mCoefL + melCoefN;                                                         // [ncep][filterN]
  int* mIdxL = reinterpret_cast<int*>(dctL + P.ncep * P.filterN);                          // [3][filterN]: start, count, offset
  for (int i = threadIdx.x; i < FFTN; i += 256) twL[i] = P.tw[i];
  for (int i = threadIdx.x; i < P.blockLen; i += 256) hamL[i] = P.ham[i];
  for (int i = threadIdx.x; i < melCoefN; i += 256) mCoefL[i] = P.mCoef[i];
  for (int i = threadIdx.x; i < P.ncep * P.filterN; i += 256) dctL[i] = P.dct[i];
  for (int i = threadIdx.x; i < P.filterN; i += 256) { mIdxL[i] = P.mStart[i]; mIdxL[P.filterN + i] = P.mCount[i]; mIdxL[2 * P.filterN + i] = P.mOff[i]; }
  __syncthreads();
  double2* bufA = bufs + (size_t) wave * 2 * N;
  double2* bufB = bufA + N;
  const int u = blockIdx.y;
  const int nsamp = nsampArr[u];
  int Tu;
  if (P.padZeros) Tu = (nsamp + P.shiftLen - 1) / P.shiftLen;
  else { long a = (long) nsamp - P.blockLen; Tu = (a > 0) ? (int) ((a + P.shiftLen - 1) / P.shiftLen) : 0; }
  const float* ys = y + (long) u * sampStride;
  const int t0 = (blockIdx.x * 4 + wave) * FW;
  for (int fi = 0; fi < FW; fi++) {
    const int t = t0 + fi;
    if (t >= Tmax) break;                                                                  // wave-uniform
    const bool live = (t < Tu);
    const long cur = (long) t * P.shiftLen;
    if (!live) {                                                                           // past the utterance's last frame: zero rows
      for (int k = lane; k < P.ncep; k += 64) cep[((long) u * Tmax + t) * P.ncep + k] = 0.0f;
      continue;
    }
    double* zr = reinterpret_cast<double*>(bufA);
    // The frame's samples come in ONCE, a coalesced run of 64 per load; the pre-emphasis' "sample before" is the neighbouring lane's (lane 0: the last lane of the
    // run before; the frame's first sample: the last sample of the frame before, feature.cc's carried prior).  One load per sample instead of two, none of them
    // behind a branch on the sample index (the framing + window stage was 1.6 of the kernel's 3.9 ms).
    // (asking for the NEXT frame's samples before this frame is worked on was measured too: 3.63 against 3.53 ms -- eight more registers a lane, nothing hidden that
    // the three waves of a SIMD did not hide already)
    constexpr int NJ = FFTN / 64;
    float smp[NJ]; float prior0 = 0.0f;
#ifndef DSR_MFCC_NOWIN
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) { const int i = 64 * jj + lane; const long n = cur + i; smp[jj] = (i < P.blockLen && n < nsamp) ? ys[n] : 0.0f; }
    if (P.preOn && t > 0) { const long n1 = (long) (t - 1) * P.shiftLen + P.blockLen - 1; prior0 = (n1 < nsamp) ? ys[n1] : 0.0f; }
#endif
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
      const int i = 64 * jj + lane;
      double v = 0.0;
#ifndef DSR_MFCC_NOWIN                                                                      /* (measurement only: no samples read, no pre-emphasis, no window) */
      float prior = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(smp[jj]), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));    // lane - 1's sample, no LDS round trip
      if (jj > 0) { const float last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(smp[jj - 1]), 63)); if (lane == 0) prior = last; } else if (lane == 0) prior = prior0;
      if (i < P.blockLen) {
        const float b = smp[jj];
        float pre = b;
        if (P.preOn) pre = (float) __dsub_rn((double) b, __dmul_rn(P.mu, (double) prior));
        const float hm = (float) __dmul_rn(hamL[i], (double) pre);
        v = (double) hm;
      }
#endif
      zr[i] = v;
    }
    stage_sync<true>();
#ifdef DSR_MFCC_NOFFT                                                                       /* (measurement only, tools/build_variants.sh: what the other stages cost) */
    double2* Z = bufA;
#else
    double2* Z = fft_lds_d<N, true>(bufA, bufB, twL, 2, -1, lane, 64);
#endif
    double* pw = reinterpret_cast<double*>(Z == bufA ? bufB : bufA);
#ifdef DSR_MFCC_NOPOW                                                                       /* (measurement only: no even/odd split, no power spectrum) */
    for (int f = lane; f < 0; f += 64) {
#else
    for (int f = lane; f < P.powN; f += 64) {
#endif
      const int ff = (f <= N) ? f : (FFTN - f);
      const double2 zf = Z[ff & (N - 1)]; double2 zc = Z[(N - ff) & (N - 1)]; zc.y = -zc.y;
      const double2 E = make_double2(0.5 * (zf.x + zc.x), 0.5 * (zf.y + zc.y));
      const double2 dd = make_double2(zf.x - zc.x, zf.y - zc.y);
      const double2 O = make_double2(0.5 * dd.y, -0.5 * dd.x);
      double2 w = twL[ff]; w.y = -w.y;
      const double re = E.x + (w.x * O.x - w.y * O.y), im = E.y + (w.x * O.y + w.y * O.x);
      pw[f] = __dadd_rn(__dmul_rn(re, re), __dmul_rn(im, im));
    }
    stage_sync<true>();
    if (powOut) for (int f = lane; f < P.powN; f += 64) powOut[((long) u * Tmax + t) * P.powN + f] = (float) pw[f];
    double* vt = pw;                                                                       // no VTLN: the mel bank reads the power spectrum where it lies
    if (P.vtlnOn) {
      vt = reinterpret_cast<double*>(Z);
      for (int k = lane; k < P.powN; k += 64) {
        const int s0 = P.vStart[k], n = P.vCount[k], o = P.vOff[k];
        double z = 0.0;
        for (int i = 0; i < n; i++) {
          double in = pw[s0 + i]; if (P.vtlnRoundFloat) in = (double) (float) in;
          z = __dadd_rn(z, __dmul_rn(P.vCoef[o + i], in));
        }
        const double dv = P.vDiv[k]; if (dv != 0.0) z = z / dv;
        vt[k] = z;
      }
      stage_sync<true>();
    }
    float* lg = reinterpret_cast<float*>(vt == pw ? reinterpret_cast<double*>(Z) : pw);    // the buffer the mel bank does not read
#ifdef DSR_MFCC_NOTAIL                                                                     /* (measurement only: no mel bank, no log, no DCT) */
    for (int k = lane; k < P.ncep; k += 64) cep[((long) u * Tmax + t) * P.ncep + k] = (float) vt[k];
    stage_sync<true>();
    continue;
#endif
    for (int j = lane; j < P.filterN; j += 64) {
      const int s0 = mIdxL[j], n = mIdxL[P.filterN + j], o = mIdxL[2 * P.filterN + j];
      double sum = 0.0; int i = 0;
      for (; i + 4 <= n; i += 4) {
        double g4 = __dmul_rn(vt[s0 + i], (double) mCoefL[o + i]);
        g4 = __dadd_rn(g4, __dmul_rn(vt[s0 + i + 1], (double) mCoefL[o + i + 1]));
        g4 = __dadd_rn(g4, __dmul_rn(vt[s0 + i + 2], (double) mCoefL[o + i + 2]));
        g4 = __dadd_rn(g4, __dmul_rn(vt[s0 + i + 3], (double) mCoefL[o + i + 3]));
        sum = __dadd_rn(sum, g4);
      }
      for (; i < n; i++) sum = __dadd_rn(sum, __dmul_rn(vt[s0 + i], (double) mCoefL[o + i]));
      double val = sum;
      if (P.sphinx) { if (val < 1.0E-05) val = 1.0E-05; }
      else { val = __dadd_rn(val, P.logA); if (val <= 0.0) val = 1.0; }
      lg[j] = (float) __dmul_rn(P.logM, log10(val));
    }
    stage_sync<true>();
    if (logmelOut) for (int j = lane; j < P.filterN; j += 64) logmelOut[((long) u * Tmax + t) * P.filterN + j] = lg[j];
    for (int k = lane; k < P.ncep; k += 64) {
      float temp = 0.0f;
      for (int j = 0; j < P.filterN; j++) temp = __fadd_rn(temp, __fmul_rn(lg[j], dctL[k * P.filterN + j]));
      cep[((long) u * Tmax + t) * P.ncep + k] = __fadd_rn(0.0f, temp);
    }
    stage_sync<true>();                                                                    // the next frame overwrites the buffers
  }
}

// mode 1: batch mean/variance (two sequential passes in fp32, as _calcMeanVariance), mode 2: run-on.
// wgt (optional): per-frame weights [U][Tmax * wStride], element 0 of the weight stream's frames (MeanSubtractionFeature(src, weight, ...), feature.cc:2577-2707)
__global__ void k_cmn(const float* __restrict__ cep, const int* __restrict__ Tarr, int U, int Tmax, int N, int mode,
                      double devNormFactor, float* __restrict__ out, const float* __restrict__ wgt, int wStride)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= U * N) return;
  const int u = idx / N, i = idx - u * N;
  const int Tu = Tarr ? Tarr[u] : Tmax;
  const int T = Tu < Tmax ? Tu : Tmax;
  const float* x = cep + (long) u * Tmax * N + i; float* o = out + (long) u * Tmax * N + i;
  if (mode == 1) {
    float m = 0.0f; double ttl = 0.0;
    const float* wu = wgt ? wgt + (long) u * Tmax * wStride : nullptr;
    for (int t = 0; t < T; t++) { const float w = wu ? wu[(long) t * wStride] : 1.0f; m = __fadd_rn(m, __fmul_rn(w, x[(long) t * N])); ttl += (double) w; }
    m = (float) ((double) m / ttl);
    float v = 0.0f;
    if (devNormFactor > 0.0) {
      for (int t = 0; t < T; t++) { const float w = wu ? wu[(long) t * wStride] : 1.0f; const float f = x[(long) t * N]; v = __fadd_rn(v, __fmul_rn(__fmul_rn(w, f), f)); }
      v = (float) __dsub_rn((double) v / ttl, (double) __fmul_rn(m, m));
    }
    for (int t = 0; t < T; t++) {
      float r = __fsub_rn(x[(long) t * N], m);
      if (devNormFactor > 0.0) { float va = v; if (va < 0.0001f) va = 0.0001f; r = (float) ((double) r / __dmul_rn(devNormFactor, (double) __fsqrt_rn(va))); }
      o[(long) t * N] = r;
    }
  } else {
    float m = 0.0f, sm = 1.0f; unsigned framesN = 0;
    const float* wu = wgt ? wgt + (long) u * Tmax * wStride : nullptr;
    for (int t = 0; t < T; t++) {
      const float f = x[(long) t * N];
      if (!wu || wu[(long) t * wStride] > 0.0f) {                      // frames without weight are normalised but leave the statistics alone (:2587)
      const float wgt = (framesN < 500) ? 0.98f : 0.995f;
      m = (float) __dadd_rn((double) __fmul_rn(wgt, m), __dmul_rn(__dsub_rn(1.0, (double) wgt), (double) f));
      if (devNormFactor > 0.0) {
        const float diff = __fsub_rn(f, m);
        sm = (float) __dadd_rn((double) __fmul_rn(wgt, sm), __dmul_rn(__dsub_rn(1.0, (double) wgt), (double) __fmul_rn(diff, diff)));
      }
      framesN++;
      }
      float r = __fsub_rn(f, m);
      if (devNormFactor > 0.0) { float va = sm; if (va < 0.0001f) va = 0.0001f; r = (float) ((double) r / __dmul_rn(devNormFactor, (double) __fsqrt_rn(va))); }
      o[(long) t * N] = r;
    }
  }
  for (int t = T; t < Tmax; t++) o[(long) t * N] = 0.0f;
}

// Batch mean/variance normalisation of one utterance per workgroup: the cepstra are read once, coalesced, into LDS; N threads run the reference's
// sequential fp32 sums from there (the thread-per-(utterance, coefficient) kernel above puts fewer wavefronts on the chip than it has CUs and
// walks memory with a stride); every thread then normalises and writes coalesced.  Same operations in the same order: the same bits.
__global__ __launch_bounds__(256) void k_cmn_lds(const float* __restrict__ cep, const int* __restrict__ Tarr, int Tmax, int N, double devNormFactor,
                                                 float* __restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  __shared__ float s_m[64], s_d[64];
  const int u = blockIdx.x;
  const int Tu = Tarr ? Tarr[u] : Tmax;
  const int T = Tu < Tmax ? Tu : Tmax;
  const float* x = cep + (long) u * Tmax * N; float* o = out + (long) u * Tmax * N;
  const int n = T * N;
  for (int j = threadIdx.x; j < n; j += 256) xs[j] = x[j];
  __syncthreads();
  if ((int) threadIdx.x < N) {
    const int i = threadIdx.x;
    float m = 0.0f; double ttl = 0.0;
    for (int t = 0; t < T; t++) { m = __fadd_rn(m, __fmul_rn(1.0f, xs[t * N + i])); ttl += 1.0; }
    m = (float) ((double) m / ttl);
    float v = 0.0f;
    if (devNormFactor > 0.0) {
      for (int t = 0; t < T; t++) { const float f = xs[t * N + i]; v = __fadd_rn(v, __fmul_rn(__fmul_rn(1.0f, f), f)); }
      v = (float) __dsub_rn((double) v / ttl, (double) __fmul_rn(m, m));
    }
    float va = v; if (va < 0.0001f) va = 0.0001f;
    s_m[i] = m; s_d[i] = va;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += 256) {
    const int i = j % N;
    float r = __fsub_rn(xs[j], s_m[i]);
    if (devNormFactor > 0.0) r = (float) ((double) r / __dmul_rn(devNormFactor, (double) __fsqrt_rn(s_d[i])));
    o[j] = r;
  }
  for (int j = n + threadIdx.x; j < Tmax * N; j += 256) o[j] = 0.0f;
}

// out[u][t][i] = sum_j A[i][j] * splice(t)[j], splice slot s = frame clamp(t+s-delta, 0, T-1)
// A workgroup owns FB consecutive frames of one utterance.  LDS holds the transform as [outDim][2 delta + 1][Np] (rows of N coefficients padded
// to Np = multiple of 4) and the FB + 2 delta (clamped) input rows, padded the same way: every LDS read is an aligned 16-byte read, the
// accumulation order (slot by slot, coefficient by coefficient, fp32, no FMA: gsl_blas_sgemv's reference loop) is unchanged.
__global__ __launch_bounds__(256) void k_splice_lda(const float* __restrict__ in, const int* __restrict__ Tarr, int Tmax, int N,
                                                    int delta, int outDim, const float* __restrict__ A, float* __restrict__ out, int FB)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int S = 2 * delta + 1, W = S * N, Np = (N + 3) & ~3;
  const int u = blockIdx.y, t0 = blockIdx.x * FB;
  int T = Tarr[u] < Tmax ? Tarr[u] : Tmax;
  if (delta > 0 && T < delta) T = 0;                 // AdjacentFeature cannot be primed (feature.cc:2861-2866)
  const float* x = in + (long) u * Tmax * N;
  const int od = A ? outDim : W;
  float* o = out + (long) u * Tmax * od;
  int t1 = t0 + FB; if (t1 > Tmax) t1 = Tmax;
  if (!A) {                                          // splice only
    for (int idx = threadIdx.x; idx < (t1 - t0) * od; idx += blockDim.x) {
      const int t = t0 + idx / od, i = idx - (t - t0) * od; float r = 0.0f;
      if (t < T) { const int s = i / N, k = i - s * N; int src = t + s - delta; if (src < 0) src = 0; if (src > T - 1) src = T - 1; r = x[(long) src * N + k]; }
      o[(long) t * od + i] = r;
    }
    return;
  }
  float* a = reinterpret_cast<float*>(smem);                         // [outDim][S][Np]
  float* xs = a + (size_t) outDim * S * Np;                          // [FB + 2 delta][Np]
  for (int i = threadIdx.x; i < outDim * S * Np; i += blockDim.x) { const int k = i % Np, q = i / Np; a[i] = (k < N) ? A[(long) (q / S) * W + (q % S) * N + k] : 0.0f; }
  if (T > 0)
    for (int i = threadIdx.x; i < (FB + 2 * delta) * Np; i += blockDim.x) {
      const int k = i % Np, rr = i / Np; int src = t0 - delta + rr; if (src < 0) src = 0; if (src > T - 1) src = T - 1;
      xs[i] = (k < N) ? x[(long) src * N + k] : 0.0f;
    }
  __syncthreads();
  for (int idx = threadIdx.x; idx < (t1 - t0) * od; idx += blockDim.x) {
    const int tl = idx / od, i = idx - tl * od, t = t0 + tl;
    float r = 0.0f;
    if (t < T) {
      float temp = 0.0f;
      for (int s = 0; s < S; s++) {
        const float* ar = a + ((size_t) i * S + s) * Np; const float* xr = xs + (size_t) (tl + s) * Np;
        for (int k4 = 0; k4 < Np; k4 += 4) {
          const float4 av = *reinterpret_cast<const float4*>(ar + k4), xv = *reinterpret_cast<const float4*>(xr + k4);
          temp = __fadd_rn(temp, __fmul_rn(xv.x, av.x));
          if (k4 + 1 < N) temp = __fadd_rn(temp, __fmul_rn(xv.y, av.y));
          if (k4 + 2 < N) temp = __fadd_rn(temp, __fmul_rn(xv.z, av.z));
          if (k4 + 3 < N) temp = __fadd_rn(temp, __fmul_rn(xv.w, av.w));
        }
      }
      r = __fadd_rn(0.0f, temp);
    }
    o[(long) t * od + i] = r;
  }
}

// The same product with register blocking: a thread owns one output coefficient for FT consecutive frames, so a row of the transform is read from LDS
// once per FT frames (the kernel above reads it once per frame and is bound by those reads), the input rows are wave-wide broadcasts, and the rows
// of the transform are pitched so that 16 lanes' 16-byte reads fall on 16 different bank groups.  A workgroup (nG groups of outDim threads) walks NB
// blocks of nG * FT frames with the transform staged once.  Order of every sum as above: the same bits.
template <int FT>
__global__ __launch_bounds__(256) void k_splice_lda_b(const float* __restrict__ in, const int* __restrict__ Tarr, int Tmax, int N, int delta, int outDim,
                                                      const float* __restrict__ A, float* __restrict__ out, int nG, int NB)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int S = 2 * delta + 1, W = S * N, Np = (N + 3) & ~3, pitch = S * Np + 4, FB = nG * FT;
  const int u = blockIdx.y;
  int T = Tarr[u] < Tmax ? Tarr[u] : Tmax;
  if (delta > 0 && T < delta) T = 0;                 // AdjacentFeature cannot be primed (feature.cc:2861-2866)
  const float* x = in + (long) u * Tmax * N;
  float* o = out + (long) u * Tmax * outDim;
  float* a = reinterpret_cast<float*>(smem);                         // [outDim][pitch]
  float* xs = a + (size_t) outDim * pitch;                           // [FB + 2 delta][Np]
  for (int i = threadIdx.x; i < outDim * S * Np; i += blockDim.x) {
    const int k = i % Np, q = i / Np, row = q / S, sl = q - row * S;
    a[(size_t) row * pitch + sl * Np + k] = (k < N) ? A[(long) row * W + sl * N + k] : 0.0f;
  }
  const int g = threadIdx.x / outDim, i = threadIdx.x - g * outDim;
  const bool worker = g < nG;
  for (int b = 0; b < NB; b++) {
    const int t0 = (blockIdx.x * NB + b) * FB;
    if (t0 >= Tmax) break;
    __syncthreads();                                                 // (the transform is in place / the rows of the block before are no longer read)
    if (T > 0)
      for (int j = threadIdx.x; j < (FB + 2 * delta) * Np; j += blockDim.x) {
        const int k = j % Np, rr = j / Np; int src = t0 - delta + rr; if (src < 0) src = 0; if (src > T - 1) src = T - 1;
        xs[j] = (k < N) ? x[(long) src * N + k] : 0.0f;
      }
    __syncthreads();
    if (!worker) continue;
    float temp[FT];
#pragma unroll
    for (int f = 0; f < FT; f++) temp[f] = 0.0f;
    if (t0 + g * FT < T) {
      const float* ar = a + (size_t) i * pitch;
      for (int s = 0; s < S; s++) {
        const float* xr = xs + (size_t) (g * FT + s) * Np;
        for (int k4 = 0; k4 < Np; k4 += 4) {
          const float4 av = *reinterpret_cast<const float4*>(ar + s * Np + k4);
          const bool h1 = k4 + 1 < N, h2 = k4 + 2 < N, h3 = k4 + 3 < N;
#pragma unroll
          for (int f = 0; f < FT; f++) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + f * Np + k4);
            float tt = __fadd_rn(temp[f], __fmul_rn(xv.x, av.x));
            if (h1) tt = __fadd_rn(tt, __fmul_rn(xv.y, av.y));
            if (h2) tt = __fadd_rn(tt, __fmul_rn(xv.z, av.z));
            if (h3) tt = __fadd_rn(tt, __fmul_rn(xv.w, av.w));
            temp[f] = tt;
          }
        }
      }
    }
#pragma unroll
    for (int f = 0; f < FT; f++) {
      const int t = t0 + g * FT + f;
      if (t < Tmax) o[(long) t * outDim + i] = (t < T) ? __fadd_rn(0.0f, temp[f]) : 0.0f;
    }
  }
}

__global__ void k_frame_counts(const int* nsampArr, int U, int blockLen, int shiftLen, int padZeros, int* Tarr)
{
  const int u = blockIdx.x * blockDim.x + threadIdx.x; if (u >= U) return;
  const int nsamp = nsampArr[u]; int Tu;
  if (padZeros) Tu = (nsamp + shiftLen - 1) / shiftLen;
  else { long a = (long) nsamp - blockLen; Tu = (a > 0) ? (int) ((a + shiftLen - 1) / shiftLen) : 0; }
  Tarr[u] = Tu;
}

// ------------------------------------------------------------------ host-side tables
static float mel_of(float hz) { return hz >= 0 ? (float) (2595.0 * log10(1.0 + (double) hz / 700.0)) : 0.0f; }
static float hertz_of(float m) { const double d = m / 2595.0; return (float) (700.0 * (pow(10.0, d) - 1.0)); }

static void build_mel(const dsr_mfcc_cfg& c, std::vector<int>& start, std::vector<int>& count, std::vector<int>& off,
                      std::vector<float>& coef, int& nReq)
{
  // MelFeature::_SparseMatrix::melScaleOrg / melScaleFF (feature.cc:1954-2090), float edge maths
  const int powN = c.powN; float up = c.up; if (up <= 0) up = c.rate / 2.0;
  const float df = c.rate / (4.0 * (powN / 2));
  const float mlow = mel_of(c.low), mup = mel_of(up);
  const float dm = (mup - mlow) / (c.filterN + 1);
  if (c.low < 0.0 || 2.0 * up > c.rate || c.low > up) throw Error(DSR_E_ERROR, "mel: something wrong with");
  for (int x = 0; x < c.filterN; x++) {
    const float left = hertz_of(x * dm + mlow), center = hertz_of((x + 1.0) * dm + mlow), right = hertz_of((x + 2.0) * dm + mlow);
    const float height = 2.0 / (right - left);
    const float slope1 = height / (center - left), slope2 = height / (center - right);
    const int st = (int) ceil(left / df), en = (int) floor(right / df);
    start.push_back(st); count.push_back(en - st + 1); off.push_back((int) coef.size()); nReq = en;
    float freq = st * df;
    for (int i = 0; i < en - st + 1; i++) {
      if (c.melVersion == 1) freq += df;
      coef.push_back(freq <= center ? slope1 * (freq - left) : slope2 * (freq - right));
      if (c.melVersion != 1) freq += df;
    }
  }
}

static void build_vtln(const dsr_mfcc_cfg& c, SparseRows& r, int& roundFloat)
{
  const int N = c.powN; const double ratio = c.vtlnRatio, edge = c.vtlnEdge;
  roundFloat = 0;
  if (c.vtlnVersion == 1) {                                        // nextOrg, feature.cc:1726-1766
    const double yedge = (edge < ratio) ? (edge / ratio) : 1.0;
    const double b = (yedge < 1.0) ? (1.0 - edge) / (1.0 - yedge) : 0;
    for (int cx = 0; cx < N; cx++) {
      const double Y0 = double(cx) / double(N), Y1 = double(cx + 1) / double(N);
      const double X0 = ((Y0 < yedge) ? (ratio * Y0) : (b * Y0 + 1.0 - b)) * N;
      const double X1 = ((Y1 < yedge) ? (ratio * Y1) : (b * Y1 + 1.0 - b)) * N;
      int L1 = int(X1); const double alpha1 = X1 - L1;
      int L0 = int(X0); const double alpha0 = int(X0) + 1 - X0;
      if (L0 >= N) L0 = N - 1;
      if (L1 > N) L1 = N;
      r.start.push_back(L0); r.off.push_back((int) r.coef.size()); r.div.push_back(0.0);
      if (L0 == L1) { r.coef.push_back(X1 - X0); r.count.push_back(1); }
      else {
        int n = 0; r.coef.push_back(alpha0); n++;
        for (int i = L0 + 1; i < L1; i++) { r.coef.push_back(1.0); n++; }
        if (L1 < N) { r.coef.push_back(alpha1); n++; }
        r.count.push_back(n);
      }
    }
  } else {                                                          // nextFF, feature.cc:1781-1838 as a gather
    roundFloat = 1;
    std::vector<std::vector<std::pair<int, double>>> rows(N); std::vector<double> aux(N, 0.0);
    float b = N * edge; float slope1 = ratio, slope2 = ratio;
    if (slope1 < 1.0) slope2 = (N - slope1 * b) / (N - b);
    for (int sIdx = 0; sIdx < N; sIdx++) {
      float s1 = sIdx - 0.5, s2 = sIdx + 0.5;
      float d1 = s1 * slope1; if (s1 > b) d1 = b * slope1 + (s1 - b) * slope2;
      float d2 = s2 * slope1; if (s2 > b) d2 = b * slope1 + (s2 - b) * slope2;
      const int i1 = int(floor(d1)), i2 = int(ceil(d2));
      if (i1 <= N - 1) {
        const double alpha = 1.0, alpha1 = (1.0 - (d1 - i1)) * alpha, alpha2 = (i2 - d2) * alpha;
        for (int j = i1; j <= i2; j++) {
          int k = j; if (k < 0) k = 0; if (k >= N) break;
          double a = alpha; if (j == i1) a = alpha1; if (j == i2) a = alpha2;
          rows[k].push_back(std::make_pair(sIdx, a)); aux[k] = aux[k] + a;
        }
      }
    }
    for (int k = 0; k < N; k++) {
      // the scatter visits sources in ascending order; a destination may be hit twice by one source
      // (clamped k) -- keep every hit, as a dense gather over [first,last] with zero fill
      r.off.push_back((int) r.coef.size());
      if (rows[k].empty()) { r.start.push_back(0); r.count.push_back(0); r.div.push_back(0.0); continue; }
      // expand duplicates: coefficient list is the ordered list of (source, a); encode as consecutive
      // sources where possible, otherwise fall back to one entry per hit via unit-length runs
      const int s0 = rows[k].front().first, sl = rows[k].back().first;
      std::vector<double> dense(sl - s0 + 1, 0.0); bool dup = false;
      for (size_t q = 0; q < rows[k].size(); q++) { double& cell = dense[rows[k][q].first - s0]; if (cell != 0.0) dup = true; cell += rows[k][q].second; }
      (void) dup;
      r.start.push_back(s0); r.count.push_back((int) dense.size());
      for (size_t q = 0; q < dense.size(); q++) r.coef.push_back(dense[q]);
      r.div.push_back(aux[k] > 1E-20 ? aux[k] : 0.0);
    }
  }
}

// ---- shared with the stream operators (ops.h)
void build_vtln_rows(int N, double ratio, double edge, int version, SparseRowsD& r)
{
  dsr_mfcc_cfg c; memset(&c, 0, sizeof(c)); c.powN = N; c.vtlnRatio = ratio; c.vtlnEdge = edge; c.vtlnVersion = version;
  SparseRows s; int rf = 0; build_vtln(c, s, rf);
  r.start = s.start; r.count = s.count; r.off = s.off; r.coef = s.coef; r.div = s.div; r.roundFloat = rf;
  if (r.coef.empty()) r.coef.push_back(0.0);
}
void build_mel_rows(int powN, float rate, float low, float up, int filterN, int version, SparseRowsF& r)
{
  dsr_mfcc_cfg c; memset(&c, 0, sizeof(c)); c.powN = powN; c.rate = rate; c.low = low; c.up = up; c.filterN = filterN; c.melVersion = version;
  build_mel(c, r.start, r.count, r.off, r.coef, r.nReq);
  if (r.coef.empty()) r.coef.push_back(0.f);
}
void build_dct(int ncep, int nmel, int type, std::vector<float>& dct)
{
  dct.assign((size_t) ncep * nmel, 0.f);
  if (type == 0) {                                          // gslmatrix.cc:115-123
    for (int k = 0; k < ncep; k++) { const double fac = k * M_PI / (double) (nmel - 1); float* q = &dct[(size_t) k * nmel];
      *q++ = 1.0; for (int l = 1; l < nmel - 1; l++) *q++ = 2.0 * cos(fac * l); *q = cos(k * M_PI); }
  } else if (type == 1) {                                   // :124-129
    for (int k = 0; k < ncep; k++) { const double fac = k * M_PI / (double) nmel; for (int l = 0; l < nmel; l++) dct[(size_t) k * nmel + l] = cos(fac * (l + 0.5)); }
  } else if (type == 2) {                                   // feature.cc:2466-2477
    for (int k = 0; k < ncep; k++) { const double deltaF = M_PI * float(k) / nmel;
      for (int f = 0; f < nmel; f++) { const double fr = deltaF * (f + 0.5); double cv = cos(fr) / nmel; if (f == 0) cv *= 0.5; dct[(size_t) k * nmel + f] = cv; } }
  } else throw Error(DSR_E_INDEX, "Unknown DCT type");
}
void op_cmn(const float* in, int T, int N, int mode, double devNormFactor, float* out, hipStream_t st, const float* wgt, int wStride)
{ if (T > 0) hipLaunchKernelGGL(k_cmn, dim3(cdiv(N, 64)), dim3(64), 0, st, in, (const int*) nullptr, 1, T, N, mode, devNormFactor, out, wgt, wStride); }

}  // namespace dsr

using namespace dsr;
struct dsr_mfcc : MfccPlan { DevBuf<int> d_T; };

extern "C" {

void dsr_mfcc_default_cfg(dsr_mfcc_cfg* c)
{
  memset(c, 0, sizeof(*c));
  c->blockLen = 320; c->shiftLen = 160; c->padZeros = 0; c->mu = 0.95; c->fftLen = 512; c->powN = 257;
  c->vtlnRatio = 1.0; c->vtlnEdge = 1.0; c->vtlnVersion = 1; c->rate = 16000.f; c->low = 0.f; c->up = 0.f;
  c->filterN = 30; c->melVersion = 1; c->logM = 1.0; c->logA = 1.0; c->sphinxFlooring = 0; c->ncep = 13; c->dctType = 1;
  c->cmnMode = 1; c->devNormFactor = 0.0; c->delta = 7; c->outDim = 39;
}

dsr_status dsr_mfcc_create(const dsr_mfcc_cfg* cfg, const float* lda, dsr_mfcc** out)
{
  return guard([&] {
    if (!cfg || !out) throw Error(DSR_E_PARAMETER, "null argument");
    const dsr_mfcc_cfg& c = *cfg;
    if (!is_pow2((unsigned) c.fftLen) || c.fftLen < 32 || c.fftLen > 4096) throw Error(DSR_E_DIMENSION, "fftLen=%d must be a power of two in [32,4096]", c.fftLen);
    if (c.blockLen > c.fftLen || c.blockLen < 2 || c.shiftLen < 1) throw Error(DSR_E_DIMENSION, "bad block/shift length");
    if (c.powN != c.fftLen && c.powN != c.fftLen / 2 + 1)       // feature.cc:1333-1335
      throw Error(DSR_E_CONSISTENCY, "Number of power coefficients %d does not match FFT length %d.", c.powN, c.fftLen);
    if (c.outDim > 0 && !lda) throw Error(DSR_E_PARAMETER, "linear transform requested without a matrix");
    require_device();
    dsr_mfcc* p = new dsr_mfcc(); p->c = c;
    std::vector<double> ham(c.blockLen);                          // feature.cc:1210-1212
    { const double temp = 2. * M_PI / (double) (c.blockLen - 1); for (int i = 0; i < c.blockLen; i++) ham[i] = 0.54 - 0.46 * cos(temp * i); }
    std::vector<double2> tw(c.fftLen);
    for (int k = 0; k < c.fftLen; k++) { const double a = 2.0 * M_PI * k / c.fftLen; tw[k] = make_double2(cos(a), sin(a)); }
    p->d_ham.upload(ham); p->d_tw.upload(tw);
    if (c.vtlnVersion != 0) {
      SparseRows r; build_vtln(c, r, p->vtlnRoundFloat);
      p->d_vStart.upload(r.start); p->d_vCount.upload(r.count); p->d_vOff.upload(r.off);
      if (r.coef.empty()) r.coef.push_back(0.0);
      p->d_vCoef.upload(r.coef); p->d_vDiv.upload(r.div);
    }
    std::vector<int> ms, mc, mo; std::vector<float> mco; int nReq = 0;
    build_mel(c, ms, mc, mo, mco, nReq);
    p->melN = nReq;
    if (c.powN < nReq) throw Error(DSR_E_CONSISTENCY, "Matrix columns differ: %d and %d.", c.powN, nReq);   // feature.cc:2111-2113
    for (size_t i = 0; i < ms.size(); i++) if (ms[i] + mc[i] > c.powN) throw Error(DSR_E_CONSISTENCY, "mel filter %zu reads past the power spectrum", i);
    if (mco.empty()) mco.push_back(0.f);
    p->d_mStart.upload(ms); p->d_mCount.upload(mc); p->d_mOff.upload(mo); p->d_mCoef.upload(mco); p->melCoefN = (int) mco.size();
    std::vector<float> dct; build_dct(c.ncep, c.filterN, c.dctType, dct);
    p->d_dct.upload(dct);
    if (c.outDim > 0) p->d_lda.upload(lda, (size_t) c.outDim * (2 * c.delta + 1) * c.ncep);
    *out = p;
  });
}
void dsr_mfcc_destroy(dsr_mfcc* p) { delete p; }

int dsr_mfcc_frames(const dsr_mfcc* p, int nsamp)
{
  const dsr_mfcc_cfg& c = p->c; int T;
  if (c.padZeros) T = (nsamp + c.shiftLen - 1) / c.shiftLen;
  else { long a = (long) nsamp - c.blockLen; T = a > 0 ? (int) ((a + c.shiftLen - 1) / c.shiftLen) : 0; }
  if (c.delta > 0 && T < c.delta) return 0;
  return T;
}
int dsr_mfcc_out_dim(const dsr_mfcc* p) { return p->c.outDim > 0 ? p->c.outDim : (2 * p->c.delta + 1) * p->c.ncep; }

dsr_status dsr_mfcc_run(dsr_mfcc* p, const float* y, const int32_t* nsamp, int U, int64_t sampStride, int Tmax, int stage,
                        float* feat, void* stream)
{
  return guard([&] {
    if (!p || !y || !nsamp || !feat) throw Error(DSR_E_PARAMETER, "null argument");
    if (U <= 0 || Tmax <= 0) return;
    if (U > 65535) throw Error(DSR_E_DIMENSION, "U must be <= 65535 per call");
    hipStream_t st = (hipStream_t) stream;
    const dsr_mfcc_cfg& c = p->c;
    const size_t nT = (size_t) U * Tmax;
    p->d_T.reserve(U);
    hipLaunchKernelGGL(k_frame_counts, dim3(cdiv(U, 256)), dim3(256), 0, st, nsamp, U, c.blockLen, c.shiftLen, c.padZeros, p->d_T.p);
    float* cepOut = (stage == 1) ? feat : (p->w_cep.reserve(nT * c.ncep), p->w_cep.p);
    MfccDev P;
    P.blockLen = c.blockLen; P.shiftLen = c.shiftLen; P.padZeros = c.padZeros; P.fftLen = c.fftLen; P.powN = c.powN;
    P.filterN = c.filterN; P.ncep = c.ncep; P.sphinx = c.sphinxFlooring; P.vtlnOn = (c.vtlnVersion != 0); P.preOn = (c.mu >= 0.0);
    P.vtlnRoundFloat = p->vtlnRoundFloat; P.mu = c.mu; P.logM = c.logM; P.logA = c.logA;
    P.ham = p->d_ham.p; P.tw = p->d_tw.p; P.vStart = p->d_vStart.p; P.vCount = p->d_vCount.p; P.vOff = p->d_vOff.p;
    P.vCoef = p->d_vCoef.p; P.vDiv = p->d_vDiv.p; P.mStart = p->d_mStart.p; P.mCount = p->d_mCount.p; P.mOff = p->d_mOff.p;
    P.mCoef = p->d_mCoef.p; P.dct = p->d_dct.p;
    const int FPB = 4;
    const size_t lds = (size_t) FPB * 2 * (c.fftLen / 2) * sizeof(double2);
    dim3 grid(cdiv(Tmax, FPB), U);
    float* powOut = stage == 4 ? feat : nullptr; float* lmOut = stage == 3 ? feat : nullptr;
#define LAUNCH(FN) { DSR_HIP(hipFuncSetAttribute((const void*) k_mfcc_frames<FN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
    hipLaunchKernelGGL(k_mfcc_frames<FN>, grid, dim3(64 * FPB), lds, st, P, y, nsamp, (long) sampStride, Tmax, cepOut, powOut, lmOut); }
    // tables in LDS + wave-private frames when the lot fits beside three more workgroups on a CU; the plain kernel otherwise (and on request)
#ifndef DSR_MFCC_FW
#define DSR_MFCC_FW 8
#endif
    constexpr int FW = DSR_MFCC_FW;
    const size_t ldsW = (size_t) c.fftLen * sizeof(double2) * (1 + 4) + (size_t) c.blockLen * sizeof(double)
                        + ((size_t) p->melCoefN + (size_t) c.ncep * c.filterN + 3 * (size_t) c.filterN) * sizeof(float);
    static const bool plainOnly = getenv("DSR_MFCC_PLAIN") != nullptr;
    if (!plainOnly && ldsW <= 52 * 1024 && (c.fftLen == 256 || c.fftLen == 512 || c.fftLen == 1024)) {
      dim3 gridW(cdiv(Tmax, 4 * FW), U);
#define LAUNCHW(FN) { DSR_HIP(hipFuncSetAttribute((const void*) k_mfcc_frames_w<FN, FW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsW)); \
    hipLaunchKernelGGL((k_mfcc_frames_w<FN, FW>), gridW, dim3(256), ldsW, st, P, p->melCoefN, y, nsamp, (long) sampStride, Tmax, cepOut, powOut, lmOut); }
      if (c.fftLen == 256) LAUNCHW(256) else if (c.fftLen == 512) LAUNCHW(512) else LAUNCHW(1024)
#undef LAUNCHW
    } else
    switch (c.fftLen) { case 32: LAUNCH(32) break; case 64: LAUNCH(64) break; case 128: LAUNCH(128) break; case 256: LAUNCH(256) break;
      case 512: LAUNCH(512) break; case 1024: LAUNCH(1024) break; case 2048: LAUNCH(2048) break; case 4096: LAUNCH(4096) break;
      default: throw Error(DSR_E_DIMENSION, "unsupported fftLen"); }
#undef LAUNCH
    DSR_HIP(hipGetLastError());
    if (stage == 1 || stage == 3 || stage == 4) return;
    float* cmnOut = cepOut;
    if (c.cmnMode != 0) {
      cmnOut = (stage == 2) ? feat : (p->w_cmn.reserve(nT * c.ncep), p->w_cmn.p);
      const size_t ldsC = sizeof(float) * (size_t) Tmax * c.ncep;
      static const bool plainCmn = getenv("DSR_CMN_PLAIN") != nullptr;
      if (c.cmnMode == 1 && c.ncep <= 64 && ldsC <= 64 * 1024 && !plainCmn) {
        DSR_HIP(hipFuncSetAttribute((const void*) k_cmn_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsC));
        hipLaunchKernelGGL(k_cmn_lds, dim3(U), dim3(256), ldsC, st, cepOut, p->d_T.p, Tmax, c.ncep, c.devNormFactor, cmnOut);
      } else
      hipLaunchKernelGGL(k_cmn, dim3(cdiv((long) U * c.ncep, 64)), dim3(64), 0, st, cepOut, p->d_T.p, U, Tmax, c.ncep, c.cmnMode, c.devNormFactor, cmnOut, (const float*) nullptr, 0);
      DSR_HIP(hipGetLastError());
    } else if (stage == 2) { DSR_HIP(hipMemcpyAsync(feat, cepOut, nT * c.ncep * sizeof(float), hipMemcpyDeviceToDevice, st)); }
    if (stage == 2) return;
    const int Np = (c.ncep + 3) & ~3, S2 = 2 * c.delta + 1;
    const int FB = getenv("DSR_LDA_FB") ? atoi(getenv("DSR_LDA_FB")) : 64;           // frames per workgroup (the transform is staged once per workgroup)
    const size_t lds2 = c.outDim > 0 ? sizeof(float) * ((size_t) c.outDim * S2 * Np + (size_t) (FB + 2 * c.delta) * Np) : 16;
    if (lds2 > 160 * 1024) throw Error(DSR_E_DIMENSION, "linear transform needs %zu bytes of LDS", lds2);
    // register-blocked product when a group of outDim threads fits the workgroup and the pitched transform fits LDS beside two more workgroups
    constexpr int FT = 8;
    const int nG = c.outDim > 0 ? 256 / c.outDim : 0;
    const size_t ldsB = c.outDim > 0 ? sizeof(float) * ((size_t) c.outDim * (S2 * Np + 4) + (size_t) (nG * FT + 2 * c.delta) * Np) : 0;
    static const bool plainLda = getenv("DSR_LDA_PLAIN") != nullptr;
    if (c.outDim > 0 && nG >= 1 && ldsB <= 52 * 1024 && !plainLda) {
      const int FBb = nG * FT; const int NB = 4;
      DSR_HIP(hipFuncSetAttribute((const void*) k_splice_lda_b<FT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsB));
      hipLaunchKernelGGL(k_splice_lda_b<FT>, dim3(cdiv(Tmax, FBb * NB), U), dim3(256), ldsB, st, cmnOut, p->d_T.p, Tmax, c.ncep, c.delta, c.outDim, p->d_lda.p, feat, nG, NB);
    } else {
      DSR_HIP(hipFuncSetAttribute((const void*) k_splice_lda, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds2));
      hipLaunchKernelGGL(k_splice_lda, dim3(cdiv(Tmax, FB), U), dim3(256), lds2, st, cmnOut, p->d_T.p, Tmax, c.ncep, c.delta, c.outDim,
                         c.outDim > 0 ? p->d_lda.p : nullptr, feat, FB);
    }
    DSR_HIP(hipGetLastError());
  });
}

}  // extern "C"
