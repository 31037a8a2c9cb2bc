// csrc/k_filterbank.hip -- oversampled uniform DFT analysis / synthesis banks on gfx950.
//
// Replaces OverSampledDFTAnalysisBank::next (btk/modulated/modulated.cc:412-452, buffers
// :400-410,461-516) and OverSampledDFTSynthesisBank::next (:586-664).  The reference keeps
// two ring buffers per operator and rebuilds an M-vector per frame; here a workgroup owns a
// tile of frames of one (utterance, channel): the sample window of the tile is staged once in
// LDS (coalesced HBM read), the polyphase sums are 4 LDS reads per output, and the length-M
// real DFT runs as a length-M/2 complex Stockham radix-4 FFT in LDS followed by the
// even/odd split.  Only bins 0..M/2 exist in HBM (the input is real).
//
// Closed forms used (checked against the oracle):
//   analysis : u_t[k] = sum_q h[k+qM] x[n_t-k-qM],  n_t = (t+laN+1)D-1,   X_t[f] = sum_k u_t[k] e^{+2 pi j fk/M}
//   synthesis: v_tau[k] = Re sum_f Y_tau[f] e^{-2 pi j fk/M} (Hermitian extension, DC/Nyquist imaginary parts dropped)
//              out_t[D-1-d] = sum_{i<R, t-R+1+i>=0} sum_q g[(M-1-k)+qM] v_{t-R+1+i+pd-Rq}[k],  k = d+iD
#include "common.h"
#include <cmath>

namespace dsr {

struct FbPlan {
  int M, m, r, R, D, synthesis, dctype, gain, pd, laN;
  std::vector<double> proto;
  DevBuf<float> d_proto;     // [m*M] taps as fp32
  DevBuf<float2> d_tw;       // tw[k] = e^{+2 pi j k / M}
};

// per-call frame bookkeeping: a whole-utterance call uses the plan's look-ahead and tail (laN, pd) and no history; a block of a longer stream
// (dsr_fb_analysis_block / dsr_fb_synthesis_block) carries the samples / subband frames that came before it
struct FbCall { int pd, laN; const float* hist; int histN; int tsMin; };

// ---------------------------------------------------------------------------------------------
// Stockham autosort FFT of `nfft` independent length-N sequences living in LDS (sequence f at
// x + f*N).  tw[k*twStep] = e^{+2 pi j k / N}.  sign=+1: e^{+...} (gsl backward), -1: forward.
// Cooperative: all `nthr` threads of the workgroup call it; returns the buffer holding the result.
template <int N>
__device__ __forceinline__ float2* fft_lds(float2* x, float2* y, const float2* tw, int twStep, int nfft,
                                            int sign, int tid, int nthr)
{
  int n = N, s = 1;
  const float sj = (float) sign;
  while (n >= 4) {
    const int m4 = n >> 2;           // sub-sequence quarter length
    const int twn = (N / n) * twStep;
    for (int idx = tid; idx < nfft * (N / 4); idx += nthr) {
      const int f = idx / (N / 4), j = idx - f * (N / 4);
      const int p = j / s, q = j - p * s;
      const float2* xi = x + f * N; float2* yo = y + f * N;
      const float2 a = xi[q + s * p], b = xi[q + s * (p + m4)], c = xi[q + s * (p + 2 * m4)], d = xi[q + s * (p + 3 * m4)];
      const float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
      const float2 bpd = make_float2(b.x + d.x, b.y + d.y), bmd = make_float2(b.x - d.x, b.y - d.y);
      // sign*j*(b-d)
      const float2 jb = make_float2(-sj * bmd.y, sj * bmd.x);
      float2 w1 = tw[p * twn], w2 = tw[2 * p * twn], w3 = tw[3 * p * twn];
      w1.y *= sj; w2.y *= sj; w3.y *= sj;
      const float2 r0 = make_float2(apc.x + bpd.x, apc.y + bpd.y);
      const float2 t1 = make_float2(amc.x + jb.x, amc.y + jb.y);
      const float2 t2 = make_float2(apc.x - bpd.x, apc.y - bpd.y);
      const float2 t3 = make_float2(amc.x - jb.x, amc.y - jb.y);
      yo[q + s * (4 * p + 0)] = r0;
      yo[q + s * (4 * p + 1)] = make_float2(t1.x * w1.x - t1.y * w1.y, t1.x * w1.y + t1.y * w1.x);
      yo[q + s * (4 * p + 2)] = make_float2(t2.x * w2.x - t2.y * w2.y, t2.x * w2.y + t2.y * w2.x);
      yo[q + s * (4 * p + 3)] = make_float2(t3.x * w3.x - t3.y * w3.y, t3.x * w3.y + t3.y * w3.x);
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
    n >>= 2; s <<= 2;
  }
  if (n == 2) {
    for (int idx = tid; idx < nfft * (N / 2); idx += nthr) {
      const int f = idx / (N / 2), q = idx - f * (N / 2);
      const float2 a = x[f * N + q], b = x[f * N + q + s];
      y[f * N + q] = make_float2(a.x + b.x, a.y + b.y);
      y[f * N + q + s] = make_float2(a.x - b.x, a.y - b.y);
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
  }
  return x;
}

// Sample n of a (stream, channel) row: the block's own samples for 0 <= n < nsamp, the carried history (the histN samples that came before
// the block, dsr_fb_analysis_block) for -histN <= n < 0, zero elsewhere (before the stream's start / after its end).
__device__ __forceinline__ float ld_sample(const float* __restrict__ xs, const float* __restrict__ hs, int histN, long n, int nsamp)
{ return (n >= 0) ? (n < nsamp ? xs[n] : 0.0f) : ((hs && n >= -(long) histN) ? hs[histN + n] : 0.0f); }

// LDS layout (dynamic): [tw: M float2][proto: m*M float][win: winLen float][bufA: FB*M float][bufB: FB*M float]
template <int M>
__global__ __launch_bounds__(256) void k_analysis(const float* __restrict__ x, const int* __restrict__ nsampArr,
                                                  const float* __restrict__ proto, const float2* __restrict__ twG,
                                                  float2* __restrict__ X, int C, long sampStride, int Tmax,
                                                  int m, int r, int pd, int laN, int gain, int TF, int FB, const float* __restrict__ hist, int histN)
{
  constexpr int N = M / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = M >> r;
  const int winLen = (TF - 1) * D + m * M;
  float2* tw = reinterpret_cast<float2*>(smem);
  float* h = reinterpret_cast<float*>(tw + M);
  float* win = h + m * M;
  float* bufA = win + ((winLen + 3) & ~3);
  float* bufB = bufA + FB * M;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int tile = blockIdx.x, c = blockIdx.y, u = blockIdx.z;
  const int t0 = tile * TF;
  const int nsamp = nsampArr[u];
  const int nblk = (nsamp + D - 1) / D;
  const int Tu = (nblk < laN) ? 0 : (nblk - laN + pd);
  const float* xs = x + ((long) u * C + c) * sampStride;
  float2* Xo = X + ((long) u * C + c) * (long) Tmax * (N + 1);

  for (int i = tid; i < M; i += nthr) tw[i] = twG[i];
  for (int i = tid; i < m * M; i += nthr) h[i] = proto[i];
  const long lo = (long) (t0 + laN + 1) * D - (long) m * M;
  const float* hs = hist ? hist + ((long) u * C + c) * histN : nullptr;
  for (int i = tid; i < winLen; i += nthr) win[i] = ld_sample(xs, hs, histN, lo + i, nsamp);
  __syncthreads();

  for (int fb0 = 0; fb0 < TF; fb0 += FB) {
    // polyphase sums -> bufA[(frame)*M + k]  (== z[n] = u[2n] + j u[2n+1] when read as float2)
    for (int idx = tid; idx < FB * M; idx += nthr) {
      const int fr = idx / M, k = idx - fr * M;
      const int base = (fb0 + fr) * D + m * M - 1 - k;
      float sum = 0.0f;
      for (int q = 0; q < m; q++) sum += h[k + q * M] * win[base - q * M];
      bufA[idx] = sum;
    }
    __syncthreads();
    float2* Z = fft_lds<N>(reinterpret_cast<float2*>(bufA), reinterpret_cast<float2*>(bufB), tw, 2, FB, +1, tid, nthr);
    // even/odd split and store bins 0..N
    const float g = (gain > 0) ? (float) gain : 1.0f;
    for (int idx = tid; idx < FB * (N + 1); idx += nthr) {
      const int fr = idx / (N + 1), f = idx - fr * (N + 1);
      const int t = t0 + fb0 + fr;
      if (t >= Tmax) continue;
      float2 out = make_float2(0.0f, 0.0f);
      if (t < Tu) {
        const float2 zf = Z[fr * N + (f & (N - 1))];
        float2 zc = Z[fr * N + ((N - f) & (N - 1))]; zc.y = -zc.y;
        const float2 E = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y + zc.y));
        const float2 dd = make_float2(zf.x - zc.x, zf.y - zc.y);
        const float2 O = make_float2(0.5f * dd.y, -0.5f * dd.x);       // -0.5j*(zf - zc)
        const float2 w = tw[f];
        out.x = (E.x + w.x * O.x - w.y * O.y) * g;
        out.y = (E.y + w.x * O.y + w.y * O.x) * g;
      }
      Xo[(long) t * (N + 1) + f] = out;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Wave-per-frame analysis bank (the fast path for M >= 128 and compile-time taps MT).
// One wavefront owns one frame at a time: lane l forms the polyphase sums of the adjacent outputs
// (2e, 2e+1), e = l + 64 rho, i.e. the packed complex point z[e]; the length-N (= M/2) DFT runs as an
// in-register radix-2 DIF: spans >= 64 pair registers of one lane, spans 32..1 exchange across lanes
// (ds_bpermute / DPP, no LDS memory); one bit-reversed pass through a 1 KB wave-private LDS strip puts
// the spectrum in natural order for the even/odd split.  No workgroup barrier after the window is staged,
// prototype taps live in registers, rows of M/2+1 complex64 leave as contiguous 8-byte-per-lane stores.
// LDS layout: [tw: M float2][win: winLen float][zb: waves x N float2]
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// lane exchange by a DPP control word (row-local: quad_perm / row_half_mirror / row_ror -- no LDS traffic)
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{ return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float bperm_f(float v, int byteAddr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byteAddr, __float_as_int(v))); }

template <int M, int MT>
__global__ __launch_bounds__(256) void k_analysis_w(const float* __restrict__ x, const int* __restrict__ nsampArr,
                                                    const float* __restrict__ proto, const float2* __restrict__ twG,
                                                    float2* __restrict__ X, int C, long sampStride, int Tmax,
                                                    int r, int pd, int laN, int gain, int TF, const float* __restrict__ hist, int histN)
{
  constexpr int N = M / 2, R = N / 64, LOGN = (N == 64 ? 6 : N == 128 ? 7 : N == 256 ? 8 : N == 512 ? 9 : 10);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = M >> r;
  const int winLen = (TF - 1) * D + MT * M;
  float2* tw = reinterpret_cast<float2*>(smem);
  float* win = reinterpret_cast<float*>(tw + M);
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwv = nthr >> 6;
  float2* zb = reinterpret_cast<float2*>(win + ((winLen + 3) & ~3)) + wave * (N + N / 8);     // skewed: slot i lives at i + (i >> 3)
  const int tile = blockIdx.x, c = blockIdx.y, u = blockIdx.z;
  const int t0 = tile * TF;
  const int nsamp = nsampArr[u];
  const int nblk = (nsamp + D - 1) / D;
  const int Tu = (nblk < laN) ? 0 : (nblk - laN + pd);
  const float* xs = x + ((long) u * C + c) * sampStride;
  float2* Xo = X + ((long) u * C + c) * (long) Tmax * (N + 1);

  for (int i = tid; i < M; i += nthr) tw[i] = twG[i];
  const long lo = (long) (t0 + laN + 1) * D - (long) MT * M;
  const float* hs = hist ? hist + ((long) u * C + c) * histN : nullptr;
  for (int i = tid; i < winLen; i += nthr) win[i] = ld_sample(xs, hs, histN, lo + i, nsamp);
  // Element <-> lane map.  The six cross-lane butterfly stages flip element-index bits 5..0; bit b is tied to the lane
  // exchange xor{32,16,8,7,2,1}: the four row-local ones are single DPP controls (row_ror:8, row_half_mirror, quad_perm),
  // so lane L holds element  el = (L0^L2) | (L1^L2)<<1 | L2<<2 | L3<<3 | L[5:4]<<4  of every 64-point group.
  const int L2b = (lane >> 2) & 1;
  const int el = ((lane ^ L2b) & 1) | ((((lane >> 1) ^ L2b) & 1) << 1) | (lane & 0x3C);
  // prototype taps of this lane's outputs: h[(2e+j) + qM]
  float hreg[R][2][MT];
#pragma unroll
  for (int rho = 0; rho < R; rho++)
#pragma unroll
    for (int q = 0; q < MT; q++) {
      const float2 hp = *reinterpret_cast<const float2*>(proto + 2 * (el + 64 * rho) + q * M);
      hreg[rho][0][q] = hp.x; hreg[rho][1][q] = hp.y;
    }
  __syncthreads();
  // per-stage constants: upper-half lanes compute (other - mine) * W_{2 span}^{el mod span}, lower-half lanes mine + other,
  // written as  (other + sg*mine) * wq  with  sg = -1/+1  and  wq = W / 1
  float2 wq[6]; float sg[6];
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const int span = 32 >> s; const bool upper = (el & span) != 0;
    wq[s] = upper ? tw[(el & (span - 1)) * (M / (2 * span))] : make_float2(1.f, 0.f);
    sg[s] = upper ? -1.f : 1.f;
  }
  float2 twf[R];
#pragma unroll
  for (int rho = 0; rho < R; rho++) twf[rho] = tw[lane + 64 * rho];
  const int ad32 = (lane ^ 32) << 2, ad16 = (lane ^ 16) << 2;
  const float g = (gain > 0) ? (float) gain : 1.0f;

  for (int tl = wave; tl < TF; tl += nwv) {
    const int t = t0 + tl;
    if (t >= Tmax) break;
    float2* row = Xo + (long) t * (N + 1);
    if (t >= Tu) {                                   // frames past the end of this utterance: zero rows
#pragma unroll
      for (int rho = 0; rho < R; rho++) row[lane + 64 * rho] = make_float2(0.f, 0.f);
      if (lane == 0) row[N] = make_float2(0.f, 0.f);
      continue;
    }
    const int base = tl * D + MT * M - 1;            // window index of sample n_t
    float2 z[R];
#pragma unroll
    for (int rho = 0; rho < R; rho++) {
      const int k0 = 2 * (el + 64 * rho);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int q = 0; q < MT; q++) {
        const float2 pr = *reinterpret_cast<const float2*>(win + (base - k0 - 1 - q * M));   // (x[n_t-k0-1-qM], x[n_t-k0-qM])
        s0 += hreg[rho][0][q] * pr.y; s1 += hreg[rho][1][q] * pr.x;
      }
      z[rho] = make_float2(s0, s1);
    }
    // in-lane stages (spans N/2 .. 64)
#pragma unroll
    for (int span = N / 2; span >= 64; span >>= 1) {
      const int sr = span / 64;
#pragma unroll
      for (int rho = 0; rho < R; rho++) {
        if (rho & sr) continue;
        const float2 a = z[rho], b = z[rho | sr];
        const float2 w = tw[((rho & (sr - 1)) * 64 + el) * (M / (2 * span))];
        z[rho] = make_float2(a.x + b.x, a.y + b.y);
        z[rho | sr] = cmulf(make_float2(a.x - b.x, a.y - b.y), w);
      }
    }
    // cross-lane stages (spans 32 .. 1)
#define DSR_STAGE(S, EXCH) _Pragma("unroll") for (int rho = 0; rho < R; rho++) { \
      const float ox = EXCH(z[rho].x), oy = EXCH(z[rho].y); \
      const float2 t = make_float2(ox + sg[S] * z[rho].x, oy + sg[S] * z[rho].y); \
      z[rho] = cmulf(t, wq[S]); }
#define X32(v) bperm_f(v, ad32)
#define X16(v) bperm_f(v, ad16)
#define X8(v) dpp_f<0x128>(v)        /* row_ror:8          : lane ^ 8 */
#define X7(v) dpp_f<0x141>(v)        /* row_half_mirror    : lane ^ 7 */
#define X2(v) dpp_f<0x4E>(v)         /* quad_perm [2,3,0,1]: lane ^ 2 */
#define X1(v) dpp_f<0xB1>(v)         /* quad_perm [1,0,3,2]: lane ^ 1 */
    DSR_STAGE(0, X32) DSR_STAGE(1, X16) DSR_STAGE(2, X8) DSR_STAGE(3, X7) DSR_STAGE(4, X2) DSR_STAGE(5, X1)
#undef DSR_STAGE
#undef X32
#undef X16
#undef X8
#undef X7
#undef X2
#undef X1
    // natural order through the wave-private strip
#pragma unroll
    for (int rho = 0; rho < R; rho++) { const int i = (int) (__brev((unsigned) (rho * 64 + el)) >> (32 - LOGN)); zb[i + (i >> 3)] = z[rho]; }
    wave_lds_sync();
#pragma unroll
    for (int rho = 0; rho < R; rho++) {
      const int f = lane + 64 * rho;
      const int fc = (N - f) & (N - 1);
      const float2 zf = zb[f + (f >> 3)]; float2 zc = zb[fc + (fc >> 3)]; zc.y = -zc.y;
      const float2 E = make_float2(0.5f * (zf.x + zc.x), 0.5f * (zf.y + zc.y));
      const float2 dd = make_float2(zf.x - zc.x, zf.y - zc.y);
      const float2 O = make_float2(0.5f * dd.y, -0.5f * dd.x);
      const float2 w = twf[rho];
      row[f] = make_float2((E.x + w.x * O.x - w.y * O.y) * g, (E.y + w.x * O.y + w.y * O.x) * g);
    }
    if (lane == 0) {                                 // bin N: X = Re(Z0) - Im(Z0)
      const float2 z0 = zb[0];
      row[N] = make_float2((z0.x - z0.y) * g, 0.f);
    }
    wave_lds_sync();
  }
}

// LDS layout: [tw: M float2][proto g: m*M float][v: NV*M float][bufA: FB*M float][bufB: FB*M float]
template <int M>
__global__ __launch_bounds__(1024) void k_synthesis(const float2* __restrict__ Y, const int* __restrict__ nframesArr,
                                                   const float* __restrict__ proto, const float2* __restrict__ twG,
                                                   float* __restrict__ y, int Tmax, long outStride,
                                                   int m, int r, int pd, int gain, int TO, int FB, const float2* __restrict__ hist, int histN, int tsMin)
{
  constexpr int N = M / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int R = 1 << r, D = M >> r;
  const int NV = TO + R * m - 1;
  float2* tw = reinterpret_cast<float2*>(smem);
  float* g = reinterpret_cast<float*>(tw + M);
  float* v = g + m * M;
  float* bufA = v + NV * M;
  float* bufB = bufA + FB * M;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int u = blockIdx.y;
  const int t0 = blockIdx.x * TO;
  const int Tu = nframesArr[u];
  const int nOut = (Tu - pd > 0) ? (Tu - pd) : 0;        // valid output blocks
  const float2* Yu = Y + (long) u * Tmax * (N + 1);
  float* yu = y + (long) u * outStride;

  for (int i = tid; i < M; i += nthr) tw[i] = twG[i];
  for (int i = tid; i < m * M; i += nthr) g[i] = proto[i];
  const int tA = t0 + pd - R * (m - 1) - (R - 1);       // first needed subband frame
  __syncthreads();

  for (int b0 = 0; b0 < NV; b0 += FB) {
    // build the packed spectrum Zc[f], f < N, of frames tA+b0 .. (+FB)
    for (int idx = tid; idx < FB * N; idx += nthr) {
      const int fr = idx / N, f = idx - fr * N;
      const int tau = tA + b0 + fr;
      float2 zc = make_float2(0.0f, 0.0f);
      // subband frame tau of this call; tau < 0: one of the histN frames carried over from the calls before (dsr_fb_synthesis_block)
      const bool inHist = hist && tau < 0 && tau >= -histN;
      if (b0 + fr < NV && ((tau >= 0 && tau < Tu) || inHist)) {
        const float2* Yr = inHist ? hist + ((long) u * histN + (histN + tau)) * (N + 1) : Yu + (long) tau * (N + 1);
        float2 gf = Yr[f];
        float2 gn = Yr[N - f];
        if (f == 0) { gf.y = 0.0f; gn.y = 0.0f; }      // G[0], G[N]: real parts only
        gn.y = -gn.y;                                    // conj(G[N-f])
        const float2 s = make_float2(gf.x + gn.x, gf.y + gn.y);
        const float2 d = make_float2(gf.x - gn.x, gf.y - gn.y);
        float2 w = tw[f]; w.y = -w.y;                    // e^{-2 pi j f/M}
        const float2 wd = make_float2(w.x * d.x - w.y * d.y, w.x * d.y + w.y * d.x);
        zc = make_float2(s.x - wd.y, s.y + wd.x);        // s + j*wd
      }
      reinterpret_cast<float2*>(bufA)[idx] = zc;
    }
    __syncthreads();
    float2* Z = fft_lds<N>(reinterpret_cast<float2*>(bufA), reinterpret_cast<float2*>(bufB), tw, 2, FB, -1, tid, nthr);
    for (int idx = tid; idx < FB * N; idx += nthr) {
      const int fr = idx / N;
      if (b0 + fr < NV) reinterpret_cast<float2*>(v)[(b0 + fr) * N + (idx - fr * N)] = Z[idx];
    }
    __syncthreads();
  }

  const float gf = (gain > 0) ? (float) gain : 1.0f;
  for (int idx = tid; idx < TO * D; idx += nthr) {
    const int tt = idx / D, d = idx - tt * D;
    const int t = t0 + tt;
    if ((long) t * D + (D - 1 - d) >= outStride) continue;
    float acc = 0.0f;
    if (t < nOut) {
      for (int i = 0; i < R; i++) {
        const int ts = t - R + 1 + i;                    // s_{ts}
        if (ts < tsMin) continue;                        // before the stream's first output block (tsMin = 0 for a whole utterance)
        const int k = d + i * D;
        float s = 0.0f;
        for (int q = 0; q < m; q++) {
          const int tau = ts + pd - R * q;               // v_{tau}[k]
          s += g[(M - 1 - k) + q * M] * v[(tau - tA) * M + k];
        }
        acc += s;
      }
      acc *= gf;
    }
    yu[(long) t * D + (D - 1 - d)] = acc;
  }
}

template <int M> static void launch_analysis(const FbPlan& p, const FbCall& k, const float* x, const int* nsamp, int U, int C,
                                             long sampStride, int Tmax, float* X, hipStream_t st)
{
  const int D = p.D;
  int FB = 4096 / M; if (FB < 1) FB = 1;
  int TF = FB * 2; if (TF > 64) TF = 64; if (TF < FB) TF = FB;
  const int winLen = (TF - 1) * D + p.m * M;
  size_t lds = sizeof(float2) * M + sizeof(float) * ((size_t) p.m * M + ((winLen + 3) & ~3) + 2 * (size_t) FB * M);
  if (lds > 160 * 1024) throw Error(DSR_E_DIMENSION, "analysis bank M=%d m=%d needs %zu bytes of LDS", M, p.m, lds);
  DSR_HIP(hipFuncSetAttribute((const void*) k_analysis<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  dim3 grid(cdiv(Tmax, TF), C, U);
  hipLaunchKernelGGL(k_analysis<M>, grid, dim3(256), lds, st, x, nsamp, p.d_proto.p, p.d_tw.p, (float2*) X, C,
                     sampStride, Tmax, p.m, p.r, k.pd, k.laN, p.gain, TF, FB, k.hist, k.histN);
  DSR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Quarter-wave analysis bank for M = 256 (N = 128 packed complex points): sixteen lanes own one frame, a wavefront
// works on four frames at once.  The length-128 DFT is split 8 x 16 so that almost all butterflies pair registers of
// one lane (every cross-lane radix-2 stage of the wave-per-frame kernel costs a full complex multiply per element on
// both partner lanes -- this kernel has a single such stage):
//   pass 1  lane l forms the polyphase sums of the points z[l + 16e], e = 0..7, and transforms them in registers
//           (8-point DFT over e), then turns them by w_128^(l k1);
//   LDS     [frame][k1][l] transposition inside a wave-private strip (no workgroup barrier);
//   pass 2  lane (k1, r) transforms the eight points l = 2m + r (8-point DFT over m); the halves r = 0, 1 of the
//           16-point DFT over l are joined by one exchange with the neighbouring lane (DPP quad_perm);
//   LDS     natural order [frame][f];
//   split   the whole wave walks one frame at a time: even/odd split of the packed transform, bins f = lane and
//           lane + 64 leave as two contiguous 512-byte stores per row.
// Per frame that is about a third of the vector instructions of the wave-per-frame kernel, which was bound by
// instruction issue (wave64 on SIMD16: 4 cycles per instruction), not by memory.
// LDS layout: [tw: M float2][win: winLen float][strip: waves x 576 float2]
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// V[k] = sum_e v[e] w8^(e k), w8 = exp(+2 pi j / 8); result in bit-reversed positions: V[k] = v[br3(k)]
__device__ __forceinline__ void fft8_pos(float2 (&v)[8])
{
  const float s = 0.70710678118654752440f;
#pragma unroll
  for (int i = 0; i < 4; i++) { const float2 a = cadd(v[i], v[i + 4]), b = csub(v[i], v[i + 4]); v[i] = a; v[i + 4] = b; }
  v[5] = make_float2((v[5].x - v[5].y) * s, (v[5].x + v[5].y) * s);                  // * w8^1
  v[6] = make_float2(-v[6].y, v[6].x);                                               // * w8^2 = j
  v[7] = make_float2((-v[7].x - v[7].y) * s, (v[7].x - v[7].y) * s);                 // * w8^3
#pragma unroll
  for (int h = 0; h < 8; h += 4) {
#pragma unroll
    for (int i = 0; i < 2; i++) { const float2 a = cadd(v[h + i], v[h + i + 2]), b = csub(v[h + i], v[h + i + 2]); v[h + i] = a; v[h + i + 2] = b; }
    v[h + 3] = make_float2(-v[h + 3].y, v[h + 3].x);                                 // * j
  }
#pragma unroll
  for (int h = 0; h < 8; h += 2) { const float2 a = cadd(v[h], v[h + 1]), b = csub(v[h], v[h + 1]); v[h] = a; v[h + 1] = b; }
}

template <int MT, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_analysis_q256(const float* __restrict__ x, const int* __restrict__ nsampArr,
                                                       const float* __restrict__ proto, const float2* __restrict__ twG,
                                                       float2* __restrict__ X, int C, long sampStride, int Tmax,
                                                       int pd, int laN, int gain, int TF, const float* __restrict__ hist, int histN)
{
  constexpr int M = 256, N = 128, D = 128;                     // r = 1
  constexpr int NQ = 1;                                        // quads of frames a wave works on side by side (independent chains to interleave)
  constexpr int SQ = 146, SK = 18, SZ = 4 * SQ;                // strip pitches: per frame / per k1 row (conflict-free both ways); strip size
  constexpr int BR[8] = {0, 4, 2, 6, 1, 5, 3, 7};
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // The window is stored with 32 floats of padding after every 128 samples: consecutive frames (the 16-lane quarters of a
  // wave) then start 160 floats apart and read from opposite halves of the 64 LDS banks.
  const int winLen = (TF - 1) * D + MT * M;                    // logical samples
  const int winPhys = winLen + 32 * ((winLen + 127) >> 7);
  float2* tw = reinterpret_cast<float2*>(smem);
  float* win = reinterpret_cast<float*>(tw + M);
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwv = nthr >> 6;
  float2* hT = reinterpret_cast<float2*>(win + ((winPhys + 3) & ~3));                      // taps as pairs: hT[n + 128 qq] = (h[2n + qq M], h[2n + 1 + qq M])
  float2* strip = hT + (M / 2) * MT + wave * (NQ * SZ);
  // A workgroup streams one (utterance, channel) row from start to end, TF frames at a time: taps and twiddles are set up
  // once per row, the samples of the next TF frames are fetched into registers while the current ones are transformed,
  // and the MT*M - D samples two tiles share stay in LDS.  (Workgroups that run at the same time work on different rows:
  // their addresses differ by the row pitches, not by a power-of-two tile pitch.)
  const int c = (int) (blockIdx.x % (unsigned) C), u = (int) (blockIdx.x / (unsigned) C);
  const int nsamp = nsampArr[u];
  const int nblk = (nsamp + D - 1) / D;
  const int Tu = (nblk < laN) ? 0 : (nblk - laN + pd);
  const float* xs = x + ((long) u * C + c) * sampStride;
  float2* Xo = X + ((long) u * C + c) * (long) Tmax * (N + 1);

  for (int i = tid; i < M; i += nthr) tw[i] = twG[i];
  const long lo0 = (long) (laN + 1) * D - (long) MT * M;       // first sample of the first tile's window (negative: zeros)
  const float* hs = hist ? hist + ((long) u * C + c) * histN : nullptr;            // carried history: only the first tile's window reaches before the block
  for (int i = tid; i < winLen; i += nthr) win[i + 32 * (i >> 7)] = ld_sample(xs, hs, histN, lo0 + i, nsamp);
  const int step = TF * D, keep = winLen - step;               // samples a tile brings in / shares with the tile before (multiples of 128)
  const int stepPhys = step + 32 * (step >> 7), keepPhys = keep + 32 * (keep >> 7);
  const bool vec = ((((uintptr_t) xs) | (uintptr_t) (sampStride * 4)) & 15) == 0;
  // PF float4 per thread bring in the next tile: step = TF D = 4 * PF * nthr samples (TF = 8 PF frames, 256 threads)
  const int l = lane & 15, q = lane >> 4;
  // prototype taps of this lane's points: h[2(l + 16e) + {0,1} + qq M]
  for (int i = tid; i < (M / 2) * MT; i += nthr) hT[i] = *reinterpret_cast<const float2*>(proto + 2 * i);   // 2 (n + 128 qq) = 2n + qq M
  __syncthreads();
  // twiddles: after pass 1  w_128^(l k1) = tw[2 l k1];  joining the halves of pass 2  w_16^(k2) = tw[16 k2] on the odd half
  // (read from the LDS table where they are used: sixteen complex constants per lane would cost registers the taps need)
  const int k1p = l >> 1, rr = l & 1;
  const int l2 = 2 * l, rr16 = rr ? 16 : 0;
  const float sgn = rr ? -1.f : 1.f;
  // the even/odd split: lane owns the adjacent bins 2 lane, 2 lane + 1 (one 16-byte store per row and lane)
  const float2 wf0 = tw[2 * lane], wf1 = tw[2 * lane + 1];
  const int fc0 = (N - 2 * lane) & (N - 1), fc1 = (N - 2 * lane - 1) & (N - 1);
  const int ic0 = fc0 + 8 * (fc0 >> 6), ic1 = fc1 + 8 * (fc1 >> 6);      // strip index of the mirror bins
  const int iz = 2 * lane + 8 * (lane >> 5);                              // strip index of bin 2 lane (bin 2 lane + 1 follows)
  const float g = (gain > 0) ? (float) gain : 1.0f;

  const int nTiles = (Tmax + TF - 1) / TF;
  for (int tile = 0; tile < nTiles; tile++) {
    const int t0 = tile * TF;
    // the next tile's new samples: in flight while this tile is transformed
    float4 pf[PF];
    const bool more = tile + 1 < nTiles;
    const long nlo = lo0 + (long) (tile + 1) * step + keep;    // their first sample index
    if (more) {
#pragma unroll
      for (int j = 0; j < PF; j++) {
        const int i4 = (j * nthr + tid) * 4; const long n = nlo + i4;
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < step) {
          if (vec && n >= 0 && n + 3 < nsamp) v4 = *reinterpret_cast<const float4*>(xs + n);
          else { v4.x = (n >= 0 && n < nsamp) ? xs[n] : 0.f; v4.y = (n + 1 >= 0 && n + 1 < nsamp) ? xs[n + 1] : 0.f;
                 v4.z = (n + 2 >= 0 && n + 2 < nsamp) ? xs[n + 2] : 0.f; v4.w = (n + 3 >= 0 && n + 3 < nsamp) ? xs[n + 3] : 0.f; }
        }
        pf[j] = v4;
      }
    }
    for (int tl = 4 * NQ * wave; tl < TF; tl += 4 * NQ * nwv) {
      if (t0 + tl >= Tmax) break;
      // ---- pass 1: polyphase sums + 8-point DFT over e + turn
      float2 v[NQ][8];
#pragma unroll
      for (int h = 0; h < NQ; h++) {
        const float* wp = win + (tl + 4 * h + q) * 160 - 2 * l;          // padded index of sample  128 (tl + q) - 2l
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int qq = 0; qq < MT; qq++) {
            const int A = MT * M - 2 - 32 * e - 256 * qq;                // sample n_t - k0 - 1 - qq M relative to the frame's first block
            const float2 pr = *reinterpret_cast<const float2*>(wp + (A + 32 * (A >> 7)));     // (x[n_t-k0-1-qM], x[n_t-k0-qM])
            const float2 hp = hT[l + 16 * e + (M / 2) * qq];
            s0 += hp.x * pr.y; s1 += hp.y * pr.x;
          }
          v[h][e] = make_float2(s0, s1);
        }
      }
#pragma unroll
      for (int h = 0; h < NQ; h++) fft8_pos(v[h]);
#pragma unroll
      for (int h = 0; h < NQ; h++)
#pragma unroll
        for (int k = 0; k < 8; k++) strip[h * SZ + q * SQ + k * SK + l] = (k == 0) ? v[h][0] : cmulf(v[h][BR[k]], tw[(l2 * k) & (M - 1)]);
      wave_lds_sync();
      // ---- pass 2: 8-point DFT over m (points l = 2m + rr), join the halves with the neighbouring lane
#pragma unroll
      for (int h = 0; h < NQ; h++)
#pragma unroll
        for (int m2 = 0; m2 < 8; m2++) v[h][m2] = strip[h * SZ + q * SQ + k1p * SK + 2 * m2 + rr];
#pragma unroll
      for (int h = 0; h < NQ; h++) fft8_pos(v[h]);
      wave_lds_sync();
#pragma unroll
      for (int h = 0; h < NQ; h++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const float2 b = (k == 0) ? v[h][0] : cmulf(v[h][BR[k]], tw[rr16 * k]);        // tw[0] = 1 on the even half
          const float ox = dpp_f<0xB1>(b.x), oy = dpp_f<0xB1>(b.y);                      // quad_perm [1,0,3,2]: lane ^ 1
          const int f = k1p + 8 * k + 64 * rr;                                           // even half: A0 + B, odd half: A0 - B
          strip[h * SZ + q * 144 + f + 8 * rr] = make_float2(ox + sgn * b.x, oy + sgn * b.y);
        }
      wave_lds_sync();
      // ---- even/odd split, one frame at a time: bins lane, lane + 64 (+ bin N by lane 0).
      // X[f] = (E + w O) g  with  E = (Z[f] + conj Z[N-f]) / 2,  O = -j (Z[f] - conj Z[N-f]) / 2  -- the halves and the gain are
      // one exact scale factor, zero for the rows past the end of the utterance.
      // (Measured and dropped, round 3: the quad's four rows packed in the strip and stored as ONE 16-byte-aligned run of 4128 bytes instead of four
      // 8-byte-aligned rows + four single-lane stores of bin N -- exact, 4.30 against 4.06 ms: the extra LDS traffic costs more than the aligned stores
      // save.  The kernel is LDS-bound: SQ_LDS_IDX_ACTIVE = 72 % of the CU-busy cycles, SQ_WAIT_INST_LDS = 18 % of the wave cycles (profiles/r03_fb_pmc.txt).)
#pragma unroll
      for (int fq = 0; fq < 4 * NQ; fq++) {
        const int t = t0 + tl + fq;
        if (t < Tmax && tl + fq < TF) {
          float2* row = Xo + (long) t * (N + 1);
          const float gh = (t < Tu) ? 0.5f * g : 0.0f;
          const float2* Z = strip + (fq >> 2) * SZ + (fq & 3) * 144;
          const float4 zz = *reinterpret_cast<const float4*>(Z + iz);
          const float2 zf0 = make_float2(zz.x, zz.y), zf1 = make_float2(zz.z, zz.w), zc0 = Z[ic0], zc1 = Z[ic1];
          const float2 s0 = make_float2(zf0.x + zc0.x, zf0.y - zc0.y), d0 = make_float2(zf0.x - zc0.x, zf0.y + zc0.y);
          const float2 s1 = make_float2(zf1.x + zc1.x, zf1.y - zc1.y), d1 = make_float2(zf1.x - zc1.x, zf1.y + zc1.y);
          float4 o4;
          o4.x = (s0.x + wf0.x * d0.y + wf0.y * d0.x) * gh; o4.y = (s0.y - wf0.x * d0.x + wf0.y * d0.y) * gh;
          o4.z = (s1.x + wf1.x * d1.y + wf1.y * d1.x) * gh; o4.w = (s1.y - wf1.x * d1.x + wf1.y * d1.y) * gh;
          typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));             // rows are 8-byte aligned: dwordx4 store, dword alignment suffices
          f4a8 ov = {o4.x, o4.y, o4.z, o4.w};
          __builtin_nontemporal_store(ov, reinterpret_cast<f4a8*>(reinterpret_cast<char*>(row) + 16 * lane));   // written once, read by the next kernel
          if (lane == 0) row[N] = make_float2((zf0.x - zf0.y) * 2.0f * gh, 0.f);        // bin N: X = Re(Z0) - Im(Z0)
        }
      }
      wave_lds_sync();
    }
    if (more) {                                                // slide the window: shared samples to the front, new ones behind
      __syncthreads();                                          // every wave is done with this tile's window
      float4 mv[2];                                             // keepPhys <= 2 * 4 * nthr (checked by the launcher)
#pragma unroll
      for (int j = 0; j < 2; j++) { const int i4 = (j * nthr + tid) * 4; mv[j] = (i4 < keepPhys) ? *reinterpret_cast<const float4*>(win + stepPhys + i4) : make_float4(0.f, 0.f, 0.f, 0.f); }
      __syncthreads();                                          // the shared samples are in registers: their old place may be overwritten
#pragma unroll
      for (int j = 0; j < 2; j++) { const int i4 = (j * nthr + tid) * 4; if (i4 < keepPhys) *reinterpret_cast<float4*>(win + i4) = mv[j]; }
#pragma unroll
      for (int j = 0; j < PF; j++) { const int i4 = (j * nthr + tid) * 4; if (i4 < step) *reinterpret_cast<float4*>(win + keepPhys + i4 + 32 * (i4 >> 7)) = pf[j]; }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same bank with the fixed-weight subband beamformer applied on the way out (SubbandDS / SubbandMVDR / SubbandGSC::next with weights that do
// not change from frame to frame: Y[t][f] = sum_c conj(w[f][c]) X_c[t][f], beamformer.cc:1137-1200,1297-1363,2583-2635): the channel snapshots
// X_c -- 2 x 10.4 GB of HBM traffic at 1000 utterances x 8 channels, written by one kernel and read back by the next -- never leave the chip.
// A workgroup owns 16 frames of one utterance (4 waves x 4 frames, the shape of k_analysis_q256) and walks the channels: a channel's window (the
// 16 frames' 2944 samples; consecutive tiles re-read the 896 they share) comes in through registers while the channel before is transformed,
// passes and even/odd split are those of k_analysis_q256 operation for operation, and what that kernel stores as row t of channel c is weighted
// and added, channel after channel in the order of k_bf_apply's loop, to accumulators of the lane's two bins (bin N: every lane, redundantly).
// The weights come channel-major, WT[c][N + 2] (the beamformer object keeps that copy), straight from memory: a 16-byte load per lane and channel, issued a whole transform
// ahead of its use (in LDS they cost the fourth workgroup per CU).
// LDS: [tw: M float2][win][hT][strip: waves x 576 float2]
template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void k_analysis_bf_q256(const float* __restrict__ x, const int* __restrict__ nsampArr,
                                                          const float* __restrict__ proto, const float2* __restrict__ twG, const float2* __restrict__ WT,
                                                          float2* __restrict__ Y, int C, long sampStride, int Tmax, int pd, int laN, int gain)
{
  constexpr int M = 256, N = 128, D = 128, TF = 4 * NW, NT = 64 * NW;
  constexpr int SQ = 146, SK = 18, SZ = 4 * SQ;
  constexpr int BR[8] = {0, 4, 2, 6, 1, 5, 3, 7};
  constexpr int winLen = (TF - 1) * D + MT * M;                // logical samples of a tile's window
  constexpr int winPhys = winLen + 32 * ((winLen + 127) >> 7);
  constexpr int NL = (winLen / 4 + NT - 1) / NT;                 // float4 per thread that bring a window in
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw = reinterpret_cast<float2*>(smem);
  float* win = reinterpret_cast<float*>(tw + M);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float2* hT = reinterpret_cast<float2*>(win + ((winPhys + 3) & ~3));
  float2* strip = hT + (M / 2) * MT + wave * SZ;
  const int u = blockIdx.y, t0 = blockIdx.x * TF;
  const int nsamp = nsampArr[u];
  const int nblk = (nsamp + D - 1) / D;
  const int Tu = (nblk < laN) ? 0 : (nblk - laN + pd);
  const long lo = (long) (laN + 1) * D - (long) MT * M + (long) t0 * D;      // first sample of this tile's window (negative: zeros)
  const bool vec = ((((uintptr_t) x) | (uintptr_t) (sampStride * 4)) & 15) == 0;

  auto fetch = [&](const int c, float4 (&pf)[NL]) __attribute__((always_inline)) {
    const float* xs = x + ((long) u * C + c) * sampStride;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const int i4 = (j * NT + tid) * 4; const long n = lo + i4;
      float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i4 < winLen) {
        if (vec && n >= 0 && n + 3 < nsamp) v4 = *reinterpret_cast<const float4*>(xs + n);
        else { v4.x = (n >= 0 && n < nsamp) ? xs[n] : 0.f; v4.y = (n + 1 >= 0 && n + 1 < nsamp) ? xs[n + 1] : 0.f;
               v4.z = (n + 2 >= 0 && n + 2 < nsamp) ? xs[n + 2] : 0.f; v4.w = (n + 3 >= 0 && n + 3 < nsamp) ? xs[n + 3] : 0.f; }
      }
      pf[j] = v4;
    }
  };
  float4 pf[NL];
  fetch(0, pf);
  for (int i = tid; i < M; i += NT) tw[i] = twG[i];
  for (int i = tid; i < (M / 2) * MT; i += NT) hT[i] = *reinterpret_cast<const float2*>(proto + 2 * i);
  __syncthreads();
  const int l = lane & 15, q = lane >> 4;
  const int k1p = l >> 1, rr = l & 1;
  const int l2 = 2 * l, rr16 = rr ? 16 : 0;
  const float sgn = rr ? -1.f : 1.f;
  const float2 wf0 = tw[2 * lane], wf1 = tw[2 * lane + 1];
  const int fc0 = (N - 2 * lane) & (N - 1), fc1 = (N - 2 * lane - 1) & (N - 1);
  const int ic0 = fc0 + 8 * (fc0 >> 6), ic1 = fc1 + 8 * (fc1 >> 6);
  const int iz = 2 * lane + 8 * (lane >> 5);
  const float g = (gain > 0) ? (float) gain : 1.0f;
  const int tl = 4 * wave;

  float4 acc[4]; float2 accN[4];
#pragma unroll
  for (int fq = 0; fq < 4; fq++) { acc[fq] = make_float4(0.f, 0.f, 0.f, 0.f); accN[fq] = make_float2(0.f, 0.f); }

  for (int c = 0; c < C; c++) {
    if (c > 0) __syncthreads();                                // every wave is done with the window of the channel before
#pragma unroll
    for (int j = 0; j < NL; j++) { const int i4 = (j * NT + tid) * 4; if (i4 < winLen) *reinterpret_cast<float4*>(win + i4 + 32 * (i4 >> 7)) = pf[j]; }
    __syncthreads();
    if (c + 1 < C) fetch(c + 1, pf);                           // in flight while this channel is transformed
    const float4 w4 = *reinterpret_cast<const float4*>(WT + c * (N + 2) + 2 * lane);         // w[2 lane][c], w[2 lane + 1][c]
    const float2 wN = WT[c * (N + 2) + N];
    if (t0 + tl < Tmax) {
      // ---- pass 1: polyphase sums + 8-point DFT over e + turn (k_analysis_q256)
      float2 v[8];
      {
        const float* wp = win + (tl + q) * 160 - 2 * l;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int qq = 0; qq < MT; qq++) {
            const int A = MT * M - 2 - 32 * e - 256 * qq;
            const float2 pr = *reinterpret_cast<const float2*>(wp + (A + 32 * (A >> 7)));
            const float2 hp = hT[l + 16 * e + (M / 2) * qq];
            s0 += hp.x * pr.y; s1 += hp.y * pr.x;
          }
          v[e] = make_float2(s0, s1);
        }
      }
      fft8_pos(v);
#pragma unroll
      for (int k = 0; k < 8; k++) strip[q * SQ + k * SK + l] = (k == 0) ? v[0] : cmulf(v[BR[k]], tw[(l2 * k) & (M - 1)]);
      wave_lds_sync();
#pragma unroll
      for (int m2 = 0; m2 < 8; m2++) v[m2] = strip[q * SQ + k1p * SK + 2 * m2 + rr];
      fft8_pos(v);
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float2 b = (k == 0) ? v[0] : cmulf(v[BR[k]], tw[rr16 * k]);
        const float ox = dpp_f<0xB1>(b.x), oy = dpp_f<0xB1>(b.y);
        const int f = k1p + 8 * k + 64 * rr;
        strip[q * 144 + f + 8 * rr] = make_float2(ox + sgn * b.x, oy + sgn * b.y);
      }
      wave_lds_sync();
      // ---- even/odd split of one frame at a time, then conj(w) x as k_bf_apply adds it
#pragma unroll
      for (int fq = 0; fq < 4; fq++) {
        const int t = t0 + tl + fq;
        const float gh = (t < Tu) ? 0.5f * g : 0.0f;
        const float2* Z = strip + fq * 144;
        const float4 zz = *reinterpret_cast<const float4*>(Z + iz);
        const float2 zf0 = make_float2(zz.x, zz.y), zf1 = make_float2(zz.z, zz.w), zc0 = Z[ic0], zc1 = Z[ic1];
        const float2 s0 = make_float2(zf0.x + zc0.x, zf0.y - zc0.y), d0 = make_float2(zf0.x - zc0.x, zf0.y + zc0.y);
        const float2 s1 = make_float2(zf1.x + zc1.x, zf1.y - zc1.y), d1 = make_float2(zf1.x - zc1.x, zf1.y + zc1.y);
        float4 o4;
        o4.x = (s0.x + wf0.x * d0.y + wf0.y * d0.x) * gh; o4.y = (s0.y - wf0.x * d0.x + wf0.y * d0.y) * gh;
        o4.z = (s1.x + wf1.x * d1.y + wf1.y * d1.x) * gh; o4.w = (s1.y - wf1.x * d1.x + wf1.y * d1.y) * gh;
        const float2 Z0 = Z[0];
        const float xN = (Z0.x - Z0.y) * 2.0f * gh;                                           // bin N: X = Re(Z0) - Im(Z0), imaginary part 0
        acc[fq].x += w4.x * o4.x + w4.y * o4.y; acc[fq].y += w4.x * o4.y - w4.y * o4.x;
        acc[fq].z += w4.z * o4.z + w4.w * o4.w; acc[fq].w += w4.z * o4.w - w4.w * o4.z;
        accN[fq].x += wN.x * xN + wN.y * 0.f; accN[fq].y += wN.x * 0.f - wN.y * xN;
      }
      wave_lds_sync();
    }
  }
  float2* Yo = Y + (long) u * Tmax * (N + 1);
#pragma unroll
  for (int fq = 0; fq < 4; fq++) {
    const int t = t0 + tl + fq;
    if (t < Tmax) {
      float2* row = Yo + (long) t * (N + 1);
      typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
      f4a8 ov = {acc[fq].x, acc[fq].y, acc[fq].z, acc[fq].w};
      *reinterpret_cast<f4a8*>(reinterpret_cast<char*>(row) + 16 * lane) = ov;
      if (lane == 0) row[N] = accN[fq];
    }
  }
}

template <int M> static void launch_synthesis(const FbPlan& p, const FbCall& k, const float* Y, const int* nframes, int U, int Tmax,
                                              long outStride, float* y, hipStream_t st)
{
  int FB = 4096 / M; if (FB < 1) FB = 1;
  int TO = 32; while (TO > 1 && (size_t) (TO + p.R * p.m - 1) * M * 4 > 48 * 1024) TO >>= 1;
  if (const char* e = getenv("DSR_SYN_TO")) { const int v = atoi(e); if (v >= 4 && v <= 64) { TO = v; } }
  if (const char* e = getenv("DSR_SYN_FB")) { const int v = atoi(e); if (v >= 1 && v <= 64) FB = v; }
  const int NV = TO + p.R * p.m - 1;
  size_t lds = sizeof(float2) * M + sizeof(float) * ((size_t) p.m * M + (size_t) NV * M + 2 * (size_t) FB * M);
  if (lds > 160 * 1024) throw Error(DSR_E_DIMENSION, "synthesis bank M=%d m=%d needs %zu bytes of LDS", M, p.m, lds);
  DSR_HIP(hipFuncSetAttribute((const void*) k_synthesis<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  const int nblkMax = (int) (outStride / p.D);
  dim3 grid(cdiv(nblkMax > 0 ? nblkMax : 1, TO), U);
  // 1024 threads: the kernel is a chain of barrier-separated LDS phases (build, four FFT stages, copy, per batch of FB frames) and two workgroups of
  // four waves per CU left it latency-bound -- 512 utterances: 1.59 ms at 256 threads, 1.08 at 512, 0.92 at 1024 (TO 32 / FB 16 stay the best shape)
  int synThr = 1024; if (const char* e = getenv("DSR_SYN_THREADS")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) synThr = v; }
  hipLaunchKernelGGL(k_synthesis<M>, grid, dim3(synThr), lds, st, (const float2*) Y, nframes, p.d_proto.p, p.d_tw.p, y,
                     Tmax, outStride, p.m, p.r, k.pd, p.gain, TO, FB, (const float2*) k.hist, k.histN, k.tsMin);
  DSR_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// NormalFFTAnalysisBank (modulated.cc:121-257) with getWindow (modulated.cc:72-97): a windowed STFT.
// Frame t transforms the M most recent samples in time order, x[(t+1)D - M + i] w[i], i = 0..M-1 (zeros before the
// start and after the end), with the forward DFT (gsl_fft_complex_radix2_forward); all M bins are kept.
// T = ceil(nsamp / D) + 1 frames (_processingDelay = 2 m - 1 = 1 zero-input frame).  One workgroup per FB frames.
template <int M>
__global__ __launch_bounds__(256) void k_normal_fft(const float* __restrict__ x, const int* __restrict__ nsampArr, const float* __restrict__ win,
                                                     const float2* __restrict__ twG, float2* __restrict__ X, int C, long sampStride,
                                                     int Tmax, int D, int FB)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw = reinterpret_cast<float2*>(smem);
  float2* bufA = tw + M; float2* bufB = bufA + (size_t) FB * M;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c = blockIdx.y, u = blockIdx.z, t0 = blockIdx.x * FB;
  const int nsamp = nsampArr[u];
  const int Tu = (nsamp + D - 1) / D + 1;
  const float* xs = x + ((long) u * C + c) * sampStride;
  float2* Xo = X + ((long) u * C + c) * (long) Tmax * M;
  for (int i = tid; i < M; i += nthr) tw[i] = twG[i];
  for (int idx = tid; idx < FB * M; idx += nthr) {
    const int fr = idx / M, i = idx - fr * M; const int t = t0 + fr;
    const long n = (long) (t + 1) * D - M + i;
    const float v = (t < Tu && n >= 0 && n < nsamp) ? xs[n] * win[i] : 0.0f;
    bufA[idx] = make_float2(v, 0.0f);
  }
  __syncthreads();
  float2* Z = fft_lds<M>(bufA, bufB, tw, 1, FB, -1, tid, nthr);
  for (int idx = tid; idx < FB * M; idx += nthr) {
    const int fr = idx / M, k = idx - fr * M; const int t = t0 + fr;
    if (t < Tmax) Xo[(long) t * M + k] = (t < Tu) ? Z[idx] : make_float2(0.f, 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// PerfectReconstructionFFTAnalysisBank / ...SynthesisBank (modulated.cc:686-970): the 2M-band cosine-modulated pair.
// Closed forms of the reference's ring buffers (checked against the literal restatement in oracle/):
//   analysis   u_t[i] = sum_k (-1)^k h[i + 2M k] x[n_t - i - (r+2) k D],  n_t = (t+1) D - 1,  i = 0..2M-1
//              X_t = (1/2M) sum_i w_i u_t[i] e^{+2 pi j f i / 2M},  w_i = e^{-j pi i / 2M};   T = ceil(nsamp/D) + 2m - 1
//   synthesis  V_t[i] = Re( (sum_f Y_t[f] e^{-2 pi j f i / 2M}) e^{+j pi i / 2M} )
//              c_t[i] = sum_k s_k g[i + 2M (m-k-1)] V_{t-(r+2)k}[i],  s_0 = +1 (m odd) / -1 (m even), alternating; t >= 2m-1
//              y_b[D-1-d] = (1/R) sum_{s<2R} c_{b+2m-1-(2R-1-s)}[d + s D],  b = 0..T-2m
// (the (r+2) k D spacing of the taps -- not 2M k for r > 0 -- is the reference's, modulated.cc:744,938.)
template <int M2>
__global__ __launch_bounds__(256) void k_pr_analysis(const float* __restrict__ x, const int* __restrict__ nsampArr, const float* __restrict__ h,
                                                      const float2* __restrict__ twG, const float2* __restrict__ wG, float2* __restrict__ X,
                                                      int C, long sampStride, int Tmax, int D, int m, int r, int FB)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw = reinterpret_cast<float2*>(smem);
  float2* bufA = tw + M2; float2* bufB = bufA + (size_t) FB * M2;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c = blockIdx.y, u = blockIdx.z, t0 = blockIdx.x * FB;
  const int nsamp = nsampArr[u];
  const int Tu = (nsamp + D - 1) / D + 2 * m - 1;
  const float* xs = x + ((long) u * C + c) * sampStride;
  float2* Xo = X + ((long) u * C + c) * (long) Tmax * M2;
  for (int i = tid; i < M2; i += nthr) tw[i] = twG[i];
  for (int idx = tid; idx < FB * M2; idx += nthr) {
    const int fr = idx / M2, i = idx - fr * M2; const int t = t0 + fr;
    float sum = 0.0f;
    if (t < Tu) {
      const long nt = (long) (t + 1) * D - 1 - i;
      float flip = 1.0f;
      for (int k = 0; k < m; k++) {
        const long n = nt - (long) (r + 2) * k * D;
        if (n >= 0 && n < nsamp) sum += flip * h[i + M2 * k] * xs[n];
        flip = -flip;
      }
    }
    const float2 w = wG[i];
    bufA[idx] = make_float2(w.x * sum, w.y * sum);
  }
  __syncthreads();
  float2* Z = fft_lds<M2>(bufA, bufB, tw, 1, FB, +1, tid, nthr);
  const float sc = 1.0f / (float) M2;
  for (int idx = tid; idx < FB * M2; idx += nthr) {
    const int fr = idx / M2, k = idx - fr * M2; const int t = t0 + fr;
    if (t < Tmax) Xo[(long) t * M2 + k] = (t < Tu) ? make_float2(Z[idx].x * sc, Z[idx].y * sc) : make_float2(0.f, 0.f);
  }
}

// synthesis, step 1: V_t[i] for every input frame
template <int M2>
__global__ __launch_bounds__(256) void k_pr_synth_fft(const float2* __restrict__ Y, const int* __restrict__ nframesArr, const float2* __restrict__ twG,
                                                       const float2* __restrict__ wG, float* __restrict__ V, int Tmax, int FB)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw = reinterpret_cast<float2*>(smem);
  float2* bufA = tw + M2; float2* bufB = bufA + (size_t) FB * M2;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int u = blockIdx.y, t0 = blockIdx.x * FB;
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const float2* Yi = Y + (long) u * Tmax * M2; float* Vo = V + (long) u * Tmax * M2;
  for (int i = tid; i < M2; i += nthr) tw[i] = twG[i];
  for (int idx = tid; idx < FB * M2; idx += nthr) { const int t = t0 + idx / M2; bufA[idx] = (t < T) ? Yi[(long) t0 * M2 + idx] : make_float2(0.f, 0.f); }
  __syncthreads();
  float2* Z = fft_lds<M2>(bufA, bufB, tw, 1, FB, -1, tid, nthr);
  for (int idx = tid; idx < FB * M2; idx += nthr) {
    const int fr = idx / M2, i = idx - fr * M2; const int t = t0 + fr;
    if (t < Tmax) { const float2 w = wG[i]; Vo[(long) t * M2 + i] = (t < T) ? Z[idx].x * w.x - Z[idx].y * w.y : 0.0f; }
  }
}
// synthesis, step 2: one thread per output sample
__global__ void k_pr_synth_out(const float* __restrict__ V, const int* __restrict__ nframesArr, const float* __restrict__ g, float* __restrict__ y,
                               int Tmax, int M2, int D, int m, int r, long outStride)
{
  const int u = blockIdx.y; const long j = (long) blockIdx.x * blockDim.x + threadIdx.x;
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const int pd = 2 * m - 1, R = 1 << r, R2 = 2 * R;
  const long nout = (T > pd) ? (long) (T - pd) * D : 0;
  if (j >= outStride) return;
  float o = 0.0f;
  if (j < nout) {
    const int b = (int) (j / D), dd = (int) (j - (long) b * D), d = D - 1 - dd;          // y_b[D-1-d]
    const float* Vu = V + (long) u * Tmax * M2;
    const float invR = 1.0f / (float) R;
    for (int s = 0; s < R2; s++) {
      const int tc = b + pd - (R2 - 1 - s);
      if (tc < pd) continue;                                     // the sample ring is still empty there
      const int i = d + s * D;
      float conv = 0.0f, flip = (m & 1) ? 1.0f : -1.0f;
      for (int k = 0; k < m; k++) {
        const int tv = tc - (r + 2) * k;
        if (tv >= 0) conv += flip * g[i + M2 * (m - k - 1)] * Vu[(long) tv * M2 + i];
        flip = -flip;
      }
      o += conv * invR;
    }
  }
  y[(long) u * outStride + j] = o;
}

#define DSR_M_DISPATCH(M_, CALL) switch (M_) { \
  case 16: CALL(16); break; case 32: CALL(32); break; case 64: CALL(64); break; case 128: CALL(128); break; \
  case 256: CALL(256); break; case 512: CALL(512); break; case 1024: CALL(1024); break; case 2048: CALL(2048); break; \
  default: throw Error(DSR_E_DIMENSION, "unsupported number of subbands M=%d (power of two in [16,2048])", M_); }

template <int M, int MT> static void launch_analysis_w(const FbPlan& p, const FbCall& k, const float* x, const int* nsamp, int U, int C,
                                                       long sampStride, int Tmax, float* X, hipStream_t st)
{
  int TF = 32; const int waves = 4;
  if (const char* e = getenv("DSR_FB_TF")) { const int v = atoi(e); if (v >= 4 && v <= 256) TF = v; }
  const int winLen = (TF - 1) * p.D + MT * M;
  size_t lds = sizeof(float2) * M + sizeof(float) * ((winLen + 3) & ~3) + sizeof(float2) * (size_t) waves * (M / 2 + M / 16);
  if (const char* e = getenv("DSR_FB_PADLDS")) lds += (size_t) atoi(e);          // occupancy experiments
  DSR_HIP(hipFuncSetAttribute((const void*) k_analysis_w<M, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  dim3 grid(cdiv(Tmax, TF), C, U);
  hipLaunchKernelGGL((k_analysis_w<M, MT>), grid, dim3(64 * waves), lds, st, x, nsamp, p.d_proto.p, p.d_tw.p, (float2*) X, C,
                     sampStride, Tmax, p.r, k.pd, k.laN, p.gain, TF, k.hist, k.histN);
  DSR_HIP(hipGetLastError());
}

template <int MT> static void launch_analysis_q256(const FbPlan& p, const FbCall& k, const float* x, const int* nsamp, int U, int C,
                                                   long sampStride, int Tmax, float* X, hipStream_t st)
{
  constexpr int M = 256;
  int TF = 16; const int waves = 4;                            // TF = 16: one pass of the workgroup (4 waves x 4 frames) per tile, 4 workgroups per CU
  if (const char* e = getenv("DSR_FB_TF")) { const int v = atoi(e); if (v == 16 || v == 32) TF = v; }
  { const int keep = MT * M - p.D, keepPhys = keep + 32 * (keep >> 7); if (keepPhys > 2 * 4 * 64 * waves || TF * p.D > (TF / 8) * 4 * 64 * waves) throw Error(DSR_E_DIMENSION, "analysis tile does not fit the streaming kernel"); }
  const int winLen = (TF - 1) * p.D + MT * M, winPhys = winLen + 32 * ((winLen + 127) >> 7);
  size_t lds = sizeof(float2) * M + sizeof(float) * ((winPhys + 3) & ~3) + sizeof(float2) * (size_t) (M / 2) * MT + sizeof(float2) * (size_t) waves * 1 * 4 * 146;
  if (const char* e = getenv("DSR_FB_PADLDS")) lds += (size_t) atoi(e);          // occupancy experiments
  dim3 grid((unsigned) U * (unsigned) C, 1, 1);
  if (TF == 16) {
    DSR_HIP(hipFuncSetAttribute((const void*) k_analysis_q256<MT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipLaunchKernelGGL((k_analysis_q256<MT, 2>), grid, dim3(64 * waves), lds, st, x, nsamp, p.d_proto.p, p.d_tw.p, (float2*) X, C, sampStride, Tmax, k.pd, k.laN, p.gain, TF, k.hist, k.histN);
  } else {
    DSR_HIP(hipFuncSetAttribute((const void*) k_analysis_q256<MT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipLaunchKernelGGL((k_analysis_q256<MT, 4>), grid, dim3(64 * waves), lds, st, x, nsamp, p.d_proto.p, p.d_tw.p, (float2*) X, C, sampStride, Tmax, k.pd, k.laN, p.gain, TF, k.hist, k.histN);
  }
  DSR_HIP(hipGetLastError());
}

void fb_analysis(const FbPlan& p, const FbCall& k, const float* x, const int* nsamp, int U, int C, long sampStride, int Tmax, float* X, hipStream_t st)
{
  if (!getenv("DSR_FB_GENERIC") && !getenv("DSR_FB_WAVE")) {
    if (p.M == 256 && p.m == 2 && p.r == 1) { launch_analysis_q256<2>(p, k, x, nsamp, U, C, sampStride, Tmax, X, st); return; }
    if (p.M == 256 && p.m == 4 && p.r == 1) { launch_analysis_q256<4>(p, k, x, nsamp, U, C, sampStride, Tmax, X, st); return; }
  }
  if (!getenv("DSR_FB_GENERIC")) {
#define W(MM, TT) if (p.M == MM && p.m == TT) { launch_analysis_w<MM, TT>(p, k, x, nsamp, U, C, sampStride, Tmax, X, st); return; }
    W(128, 2) W(128, 4) W(256, 2) W(256, 4) W(512, 2) W(512, 4) W(1024, 2) W(1024, 4)
#undef W
  }
#define CALL(MM) launch_analysis<MM>(p, k, x, nsamp, U, C, sampStride, Tmax, X, st)
  DSR_M_DISPATCH(p.M, CALL)
#undef CALL
}
template <int M> static void launch_normal_fft(const float* x, const int* nsamp, const float* win, const float2* tw, int U, int C, long sampStride,
                                               int Tmax, int D, float* X, hipStream_t st)
{
  int FB = 2048 / M; if (FB < 1) FB = 1;
  const size_t lds = sizeof(float2) * ((size_t) M + 2 * (size_t) FB * M);
  DSR_HIP(hipFuncSetAttribute((const void*) k_normal_fft<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  dim3 grid(cdiv(Tmax, FB), C, U);
  hipLaunchKernelGGL(k_normal_fft<M>, grid, dim3(256), lds, st, x, nsamp, win, tw, (float2*) X, C, sampStride, Tmax, D, FB);
  DSR_HIP(hipGetLastError());
}
void fb_normal_fft(int M, const float* x, const int* nsamp, const float* win, const float2* tw, int U, int C, long sampStride, int Tmax, int D,
                   float* X, hipStream_t st)
{
#define CALL(MM) launch_normal_fft<MM>(x, nsamp, win, tw, U, C, sampStride, Tmax, D, X, st)
  DSR_M_DISPATCH(M, CALL)
#undef CALL
}
struct PrPlan { int M, m, r, D; DevBuf<float> h; DevBuf<float2> tw, wA, wS; DevBuf<float> V; };
template <int M2> static void launch_pr_analysis(const PrPlan& p, const float* x, const int* nsamp, int U, int C, long sampStride, int Tmax, float* X, hipStream_t st)
{
  int FB = 2048 / M2; if (FB < 1) FB = 1;
  const size_t lds = sizeof(float2) * ((size_t) M2 + 2 * (size_t) FB * M2);
  DSR_HIP(hipFuncSetAttribute((const void*) k_pr_analysis<M2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  hipLaunchKernelGGL(k_pr_analysis<M2>, dim3(cdiv(Tmax, FB), C, U), dim3(256), lds, st, x, nsamp, p.h.p, p.tw.p, p.wA.p, (float2*) X, C, sampStride, Tmax, p.D, p.m, p.r, FB);
  DSR_HIP(hipGetLastError());
}
template <int M2> static void launch_pr_synth_fft(PrPlan& p, const float* Y, const int* nframes, int U, int Tmax, hipStream_t st)
{
  int FB = 2048 / M2; if (FB < 1) FB = 1;
  const size_t lds = sizeof(float2) * ((size_t) M2 + 2 * (size_t) FB * M2);
  DSR_HIP(hipFuncSetAttribute((const void*) k_pr_synth_fft<M2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  hipLaunchKernelGGL(k_pr_synth_fft<M2>, dim3(cdiv(Tmax, FB), U), dim3(256), lds, st, (const float2*) Y, nframes, p.tw.p, p.wS.p, p.V.p, Tmax, FB);
  DSR_HIP(hipGetLastError());
}
void fb_synthesis(const FbPlan& p, const FbCall& k, const float* Y, const int* nframes, int U, int Tmax, long outStride, float* y, hipStream_t st)
{
#define CALL(MM) launch_synthesis<MM>(p, k, Y, nframes, U, Tmax, outStride, y, st)
  DSR_M_DISPATCH(p.M, CALL)
#undef CALL
}

// carried history of a stream after a block: the last H items of (old history ++ the block's n valid items), item = one sample of a
// (stream, channel) row or one subband frame of rowLen floats; written to the other buffer of the pair (a call reads one and writes the other)
__global__ void k_hist_update(const float* __restrict__ oldH, float* __restrict__ newH, const float* __restrict__ blk, const int* __restrict__ nArr,
                              int rowsPerStream, long blkRowStride, int H, int rowLen, int haveOld, int nUnit /* items per count unit */)
{
  const long row = blockIdx.y;                                   // (stream, channel) row
  const int u = (int) (row / rowsPerStream);
  const long nItems = (long) nArr[u] * nUnit;
  const long total = (long) H * rowLen;
  for (long i = (long) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long) gridDim.x * blockDim.x) {
    const long it = i / rowLen, w = i - it * rowLen;             // item index in the new history, word inside the item
    const long src = nItems - H + it;                            // item index in the block (negative: still in the old history)
    float v = 0.0f;
    if (src >= 0) v = blk[row * blkRowStride + src * rowLen + w];
    else if (haveOld && H + src >= 0) v = oldH[row * total + (H + src) * rowLen + w];
    newH[row * total + i] = v;
  }
}

}  // namespace dsr

using namespace dsr;
struct dsr_fb : FbPlan {};
// state a filter bank carries from one block of a stream to the next (dsr_fb_analysis_block / dsr_fb_synthesis_block)
struct dsr_fb_state { int U = 0, C = 0, H = 0, rowLen = 1; bool synthesis = false, started = false; long emitted = 0; int cur = 0; DevBuf<float> hist[2]; };

extern "C" {

// PerfectReconstructionFFTAnalysisBank / SynthesisBank (modulated.cc:686-970)
struct dsr_prfb : PrPlan {};
dsr_status dsr_prfb_create(const double* prototype, int M, int m, int r, dsr_prfb** out)
{
  return guard([&] {
    if (!out || !prototype) throw Error(DSR_E_PARAMETER, "null argument");
    if (!is_pow2((unsigned) M) || M < 8 || M > 1024) throw Error(DSR_E_DIMENSION, "M=%d must be a power of two in [8,1024]", M);
    if (m < 1 || r < 0 || (M >> r) < 1) throw Error(DSR_E_DIMENSION, "bad m=%d r=%d", m, r);
    require_device();
    dsr_prfb* p = new dsr_prfb(); p->M = M; p->m = m; p->r = r; p->D = M >> r;
    const int M2 = 2 * M;
    std::vector<float> h((size_t) M2 * m); for (size_t i = 0; i < h.size(); i++) h[i] = (float) prototype[i];
    std::vector<float2> tw(M2), wA(M2), wS(M2);
    for (int i = 0; i < M2; i++) {
      const double a = 2.0 * M_PI * (double) i / (double) M2, b = M_PI * (double) i / (double) M2;
      tw[i] = make_float2((float) cos(a), (float) sin(a)); wA[i] = make_float2((float) cos(b), (float) -sin(b)); wS[i] = make_float2((float) cos(b), (float) sin(b));
    }
    p->h.upload(h); p->tw.upload(tw); p->wA.upload(wA); p->wS.upload(wS);
    *out = p;
  });
}
void dsr_prfb_destroy(dsr_prfb* p) { delete p; }
int dsr_prfb_fft_len(const dsr_prfb* p) { return p ? 2 * p->M : 0; }
int dsr_prfb_block_len(const dsr_prfb* p) { return p ? p->D : 0; }
int dsr_prfb_analysis_frames(const dsr_prfb* p, int nsamp) { return p ? (nsamp + p->D - 1) / p->D + 2 * p->m - 1 : 0; }
int dsr_prfb_synthesis_blocks(const dsr_prfb* p, int nframes) { return (p && nframes > 2 * p->m - 1) ? nframes - (2 * p->m - 1) : 0; }
dsr_status dsr_prfb_analysis(const dsr_prfb* p, const float* x, const int32_t* nsamp_dev, int U, int C, int64_t sampStride, int Tmax, float* X, void* stream)
{
  return guard([&] {
    if (!p || !x || !nsamp_dev || !X) throw Error(DSR_E_PARAMETER, "null argument");
    if (U <= 0 || C <= 0 || Tmax <= 0) return;
    hipStream_t st = (hipStream_t) stream;
#define CALL(MM) launch_pr_analysis<MM>(*p, x, nsamp_dev, U, C, sampStride, Tmax, X, st)
    DSR_M_DISPATCH(2 * p->M, CALL)
#undef CALL
  });
}
dsr_status dsr_prfb_synthesis(dsr_prfb* p, const float* Y, const int32_t* nframes_dev, int U, int Tmax, int64_t outStride, float* y, void* stream)
{
  return guard([&] {
    if (!p || !Y || !nframes_dev || !y) throw Error(DSR_E_PARAMETER, "null argument");
    if (U <= 0 || Tmax <= 0 || outStride <= 0) return;
    hipStream_t st = (hipStream_t) stream;
    p->V.reserve((size_t) U * Tmax * 2 * p->M);
#define CALL(MM) launch_pr_synth_fft<MM>(*p, Y, nframes_dev, U, Tmax, st)
    DSR_M_DISPATCH(2 * p->M, CALL)
#undef CALL
    hipLaunchKernelGGL(k_pr_synth_out, dim3((unsigned) ((outStride + 255) / 256), U), dim3(256), 0, st, p->V.p, nframes_dev, p->h.p, y, Tmax, 2 * p->M, p->D, p->m,
                       p->r, (long) outStride);
    DSR_HIP(hipGetLastError());
  });
}

// NormalFFTAnalysisBank (modulated.cc:121-257): windowType 0 rectangle, 1 Hamming (default), 2 Hanning (getWindow, :72-97)
struct dsr_stft { int M, r, D, winType; DevBuf<float> win; DevBuf<float2> tw; };
dsr_status dsr_stft_create(int M, int r, int windowType, dsr_stft** out)
{
  return guard([&] {
    if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (!is_pow2((unsigned) M) || M < 16 || M > 2048) throw Error(DSR_E_DIMENSION, "M=%d must be a power of two in [16,2048]", M);
    if (r < 0 || (M >> r) < 1) throw Error(DSR_E_DIMENSION, "bad r=%d", r);
    require_device();
    dsr_stft* p = new dsr_stft(); p->M = M; p->r = r; p->D = M >> r; p->winType = windowType;
    std::vector<float> w(M); std::vector<float2> tw(M);
    for (int i = 0; i < M; i++) {
      double v;
      switch (windowType) {
      case 0: v = 1.0; break;
      case 2: v = 0.5 * (1 - cos((2.0 * M_PI * i) / (double) (M - 1))); break;
      default: { const double temp = 2. * M_PI / (double) (M - 1); v = 0.54 - 0.46 * cos(temp * i); } break;
      }
      w[i] = (float) v;
      const double a = 2.0 * M_PI * (double) i / (double) M; tw[i] = make_float2((float) cos(a), (float) sin(a));
    }
    p->win.upload(w); p->tw.upload(tw);
    *out = p;
  });
}
void dsr_stft_destroy(dsr_stft* p) { delete p; }
int dsr_stft_frames(const dsr_stft* p, int nsamp) { return p ? (nsamp + p->D - 1) / p->D + 1 : 0; }
int dsr_stft_block_len(const dsr_stft* p) { return p ? p->D : 0; }
dsr_status dsr_stft_analysis(const dsr_stft* p, const float* x, const int32_t* nsamp_dev, int U, int C, int64_t sampStride, int Tmax,
                             float* X, void* stream)
{
  return guard([&] {
    if (!p || !x || !nsamp_dev || !X) throw Error(DSR_E_PARAMETER, "null argument");
    if (U <= 0 || C <= 0 || Tmax <= 0) return;
    fb_normal_fft(p->M, x, nsamp_dev, p->win.p, p->tw.p, U, C, sampStride, Tmax, p->D, X, (hipStream_t) stream);
  });
}

dsr_status dsr_fb_create(const double* prototype, int M, int m, int r, int synthesis, int dctype, int gain, dsr_fb** out)
{
  return guard([&] {
    if (!out || !prototype) throw Error(DSR_E_PARAMETER, "null argument");
    if (!is_pow2((unsigned) M) || M < 16 || M > 2048) throw Error(DSR_E_DIMENSION, "M=%d must be a power of two in [16,2048]", M);
    if (m < 1 || r < 0 || (M >> r) < 1) throw Error(DSR_E_DIMENSION, "bad m=%d r=%d", m, r);
    require_device();
    dsr_fb* p = new dsr_fb();
    p->M = M; p->m = m; p->r = r; p->R = 1 << r; p->D = M >> r; p->synthesis = synthesis; p->dctype = dctype; p->gain = gain;
    // modulated.cc:279-296
    p->laN = 0;
    switch (dctype) {
    case 1: p->pd = m * p->R - 1; break;
    case 2: if (synthesis) p->pd = m * p->R / 2; else { p->pd = m * p->R - 1; p->laN = m * p->R / 2 - 1; } break;
    default: p->pd = 2 * m - 1; break;
    }
    p->proto.assign(prototype, prototype + (size_t) m * M);
    std::vector<float> pf((size_t) m * M); for (size_t i = 0; i < pf.size(); i++) pf[i] = (float) prototype[i];
    std::vector<float2> tw(M);
    for (int k = 0; k < M; k++) { double a = 2.0 * M_PI * k / M; tw[k] = make_float2((float) cos(a), (float) sin(a)); }
    p->d_proto.upload(pf); p->d_tw.upload(tw);
    *out = p;
  });
}
void dsr_fb_destroy(dsr_fb* p) { delete p; }
int dsr_fb_analysis_frames(const dsr_fb* p, int nsamp)
{ int nblk = (nsamp + p->D - 1) / p->D; return nblk < p->laN ? 0 : nblk - p->laN + p->pd; }
int dsr_fb_synthesis_blocks(const dsr_fb* p, int nframes) { return nframes - p->pd > 0 ? nframes - p->pd : 0; }
int dsr_fb_processing_delay(const dsr_fb* p) { return p->pd; }
int dsr_fb_block_len(const dsr_fb* p) { return p->D; }

dsr_status dsr_fb_analysis(const dsr_fb* p, const float* x, const int32_t* nsamp, int U, int C, int64_t sampStride,
                           int Tmax, float* X, void* stream)
{
  return guard([&] {
    if (!p || !x || !nsamp || !X) throw Error(DSR_E_PARAMETER, "null argument");
    if (p->synthesis) throw Error(DSR_E_CONSISTENCY, "plan was created for synthesis");
    if (U <= 0 || C <= 0 || Tmax <= 0) return;
    if (C > 65535 || U > 65535) throw Error(DSR_E_DIMENSION, "U and C must be <= 65535 per call");
    const FbCall k = { p->pd, p->laN, nullptr, 0, 0 };
    fb_analysis(*p, k, x, nsamp, U, C, (long) sampStride, Tmax, X, (hipStream_t) stream);
  });
}
namespace dsr { const float2* bf_fixed_weights_dev(dsr_bf* s); }
int dsr_fb_analysis_beamform_supported(const dsr_fb* p, const dsr_bf* bf)
{
  if (!p || !bf || p->synthesis || getenv("DSR_FB_GENERIC") || getenv("DSR_FB_WAVE") || getenv("DSR_FB_NOFUSE")) return 0;
  if (!(p->M == 256 && p->r == 1 && (p->m == 2 || p->m == 4))) return 0;
  const int C = dsr_bf_chan_n(bf);
  return (dsr_bf_fft_len(bf) == p->M && !dsr_bf_half_band_shift(bf) && !dsr_bf_is_adaptive(bf) && C >= 1 && C <= 16) ? 1 : 0;
}
dsr_status dsr_fb_analysis_beamform(const dsr_fb* p, dsr_bf* bf, const float* x, const int32_t* nsamp, int U, int C, int64_t sampStride,
                                    int Tmax, float* Y, void* stream)
{
  return guard([&] {
    if (!p || !bf || !x || !nsamp || !Y) throw Error(DSR_E_PARAMETER, "null argument");
    if (!dsr_fb_analysis_beamform_supported(p, bf)) throw Error(DSR_E_PARAMETER, "analysis + beamformer in one pass needs M = 256, r = 1, m in {2, 4}, at most 16 channels and fixed weights");
    if (C != dsr_bf_chan_n(bf)) throw Error(DSR_E_DIMENSION, "beamformer has %d channels, input has %d", dsr_bf_chan_n(bf), C);
    if (U <= 0 || Tmax <= 0) return;
    if (U > 65535) throw Error(DSR_E_DIMENSION, "U must be <= 65535 per call");
    const float2* W = bf_fixed_weights_dev(bf);
    if (!W) throw Error(DSR_E_CONSISTENCY, "the beamformer has no fixed weights");
    hipStream_t st = (hipStream_t) stream;
    constexpr int M = 256;
    const int MT = p->m;
    int NW = 4; if (const char* e = getenv("DSR_FB_FUSED_WAVES")) { const int v = atoi(e); if (v == 4 || v == 8) NW = v; }
    const int TF = 4 * NW;
    const int winLen = (TF - 1) * p->D + MT * M, winPhys = winLen + 32 * ((winLen + 127) >> 7);
    const size_t lds = sizeof(float2) * M + sizeof(float) * ((winPhys + 3) & ~3) + sizeof(float2) * (size_t) (M / 2) * MT + sizeof(float2) * (size_t) NW * 4 * 146;
    dim3 grid((unsigned) cdiv(Tmax, TF), (unsigned) U);
#define LFB(MT_, NW_) { DSR_HIP(hipFuncSetAttribute((const void*) k_analysis_bf_q256<MT_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
      hipLaunchKernelGGL((k_analysis_bf_q256<MT_, NW_>), grid, dim3(64 * NW_), lds, st, x, nsamp, p->d_proto.p, p->d_tw.p, W, (float2*) Y, C, (long) sampStride, Tmax, p->pd, p->laN, p->gain); }
    if (MT == 4) { if (NW == 8) LFB(4, 8) else LFB(4, 4) } else { if (NW == 8) LFB(2, 8) else LFB(2, 4) }
#undef LFB
    DSR_HIP(hipGetLastError());
  });
}
dsr_status dsr_fb_synthesis(const dsr_fb* p, const float* Y, const int32_t* nframes, int U, int Tmax,
                            int64_t outStride, float* y, void* stream)
{
  return guard([&] {
    if (!p || !Y || !nframes || !y) throw Error(DSR_E_PARAMETER, "null argument");
    if (!p->synthesis) throw Error(DSR_E_CONSISTENCY, "plan was created for analysis");
    if (U <= 0 || Tmax <= 0 || outStride <= 0) return;
    if (U > 65535) throw Error(DSR_E_DIMENSION, "U must be <= 65535 per call");
    const FbCall k = { p->pd, 0, nullptr, 0, 0 };
    fb_synthesis(*p, k, Y, nframes, U, Tmax, (long) outStride, y, (hipStream_t) stream);
  });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Block-wise processing of long streams (BASELINE configs[4]: 10-minute streams in 10-second blocks).  The reference operators keep ring buffers
// of the last m*M samples (analysis, modulated.h:79-163, modulated.cc:400-452) and of the last R*m subband frames (synthesis, :586-664); here
// a state object carries exactly those between calls: m*M - D samples per (stream, channel), R*m - 1 subband frames per stream.
dsr_status dsr_fb_state_create(const dsr_fb* p, int U, int C, dsr_fb_state** out)
{
  return guard([&] {
    if (!p || !out || U < 1 || (!p->synthesis && C < 1)) throw Error(DSR_E_PARAMETER, "bad argument");
    require_device();
    dsr_fb_state* s = new dsr_fb_state(); s->U = U; s->synthesis = p->synthesis != 0;
    if (p->synthesis) { s->C = 1; s->H = p->R * p->m - 1; s->rowLen = 2 * (p->M / 2 + 1); }
    else { s->C = C; s->H = p->m * p->M - p->D; s->rowLen = 1; }
    const size_t n = (size_t) U * s->C * (s->H > 0 ? s->H : 1) * s->rowLen;
    s->hist[0].reserve(n); s->hist[1].reserve(n);
    *out = s;
  });
}
void dsr_fb_state_destroy(dsr_fb_state* s) { delete s; }
dsr_status dsr_fb_state_reset(dsr_fb_state* s)
{ return guard([&] { if (!s) throw Error(DSR_E_PARAMETER, "null argument"); s->started = false; s->emitted = 0; }); }

// frames a block of nsampBlock new samples yields: the stream's first block spends the look-ahead (laN source blocks, delayCompensationType 2,
// modulated.cc:467-474), its last one adds the processingDelay zero-input frames (:493-501)
int dsr_fb_analysis_block_frames(const dsr_fb* p, const dsr_fb_state* s, int nsampBlock, int last)
{
  if (!p || !s) return 0;
  const int nblk = (nsampBlock + p->D - 1) / p->D, la = s->started ? 0 : p->laN;
  return nblk < la ? 0 : nblk - la + (last ? p->pd : 0);
}
dsr_status dsr_fb_analysis_block(const dsr_fb* p, dsr_fb_state* s, const float* x, const int32_t* nsamp_dev, int U, int C, int64_t sampStride,
                                 int last, int Tmax, float* X, void* stream)
{
  return guard([&] {
    if (!p || !s || !x || !nsamp_dev || !X) throw Error(DSR_E_PARAMETER, "null argument");
    if (p->synthesis || s->synthesis) throw Error(DSR_E_CONSISTENCY, "plan / state were created for synthesis");
    if (U != s->U || C != s->C) throw Error(DSR_E_DIMENSION, "state holds %d x %d rows, call has %d x %d", s->U, s->C, U, C);
    if (s->H != p->m * p->M - p->D) throw Error(DSR_E_CONSISTENCY, "state belongs to another filter bank");
    if (Tmax <= 0) return;
    hipStream_t st = (hipStream_t) stream;
    const FbCall k = { last ? p->pd : 0, s->started ? 0 : p->laN, s->started ? s->hist[s->cur].p : nullptr, s->H, 0 };
    fb_analysis(*p, k, x, nsamp_dev, U, C, (long) sampStride, Tmax, X, st);
    if (s->H > 0) {
      int gx = cdiv(s->H, 256); if (gx < 1) gx = 1;
      hipLaunchKernelGGL(k_hist_update, dim3(gx, (unsigned) U * C), dim3(256), 0, st, s->hist[s->cur].p, s->hist[s->cur ^ 1].p, x, nsamp_dev, C, (long) sampStride,
                         s->H, 1, s->started ? 1 : 0, 1);
      DSR_HIP(hipGetLastError());
      s->cur ^= 1;
    }
    s->started = true;
  });
}
// output blocks a call with nframesBlock new subband frames yields: the first call keeps processingDelay frames of look-ahead back (modulated.cc:631-634)
int dsr_fb_synthesis_block_blocks(const dsr_fb* p, const dsr_fb_state* s, int nframesBlock)
{
  if (!p || !s) return 0;
  if (s->started) return nframesBlock;
  return nframesBlock - p->pd > 0 ? nframesBlock - p->pd : 0;
}
dsr_status dsr_fb_synthesis_block(const dsr_fb* p, dsr_fb_state* s, const float* Y, const int32_t* nframes_dev, int nframesHostMax, int U, int Tmax,
                                  int64_t outStride, float* y, void* stream)
{
  return guard([&] {
    if (!p || !s || !Y || !nframes_dev || !y) throw Error(DSR_E_PARAMETER, "null argument");
    if (!p->synthesis || !s->synthesis) throw Error(DSR_E_CONSISTENCY, "plan / state were created for analysis");
    if (U != s->U) throw Error(DSR_E_DIMENSION, "state holds %d streams, call has %d", s->U, U);
    if (s->H != p->R * p->m - 1) throw Error(DSR_E_CONSISTENCY, "state belongs to another filter bank");
    if (Tmax <= 0 || outStride <= 0) return;
    if (!s->started && nframesHostMax < p->pd + p->R * p->m) throw Error(DSR_E_DIMENSION, "the first block of a stream must hold at least %d subband frames", p->pd + p->R * p->m);
    hipStream_t st = (hipStream_t) stream;
    // later calls: local frame tau = absolute frame minus the frames of the calls before; the look-ahead is already inside the history shift
    const FbCall k = { s->started ? 0 : p->pd, 0, s->started ? s->hist[s->cur].p : nullptr, s->H, s->started ? -(p->R - 1) : 0 };
    fb_synthesis(*p, k, Y, nframes_dev, U, Tmax, (long) outStride, y, st);
    if (s->H > 0) {
      int gx = cdiv((long) s->H * s->rowLen, 256); if (gx < 1) gx = 1; if (gx > 64) gx = 64;
      hipLaunchKernelGGL(k_hist_update, dim3(gx, (unsigned) U), dim3(256), 0, st, s->hist[s->cur].p, s->hist[s->cur ^ 1].p, Y, nframes_dev, 1, (long) Tmax * s->rowLen,
                         s->H, s->rowLen, s->started ? 1 : 0, 1);
      DSR_HIP(hipGetLastError());
      s->cur ^= 1;
    }
    s->started = true;
  });
}

}  // extern "C"
