// csrc/k_gmm_mfma.hip -- mode 2 of dsr_gmm_score: nearest-Gaussian scoring as a frame x Gaussian contraction on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32), the one dense GEMM-shaped stage of the path.
//
// Replaces the per-(codebook, frame) scalar loop of CodebookBasic::_scoreOpt (asr/gaussian/codebookBasic.cc:509-535)
// by the expanded quadratic   dist_j(x) = [pi + det_j + sum mu^2 iv] + sum_d iv_jd x_d^2 - 2 mu_jd iv_jd x_d,
// i.e. D = A B with  A[j] = (iv_j, -2 mu_j iv_j, const_j, 0)  (Gaussians as MFMA rows) and
// B[:, n] = (x_n^2, x_n, 1, 0)  (frames as MFMA columns); K' = 2*dimN+1 padded to even (80 for 39-dim).
// A 256-thread workgroup owns 128 frames (one 32-frame column tile per wavefront, B fragments in registers) and
// streams the Gaussian table in 32-row chunks through LDS; the 32x128 tile of distances goes to LDS and one thread
// per frame walks its rows, keeping the running per-codebook minimum (strict '<': first Gaussian wins, as the
// reference); scores are staged in LDS and written in contiguous runs.
// Numerics: fp32 fmaf chain in k order (exact-f32 MFMA); the expanded form differs from the reference's
// (mu-x)^2*iv accumulation by ~1e-6 relative -- this mode carries the stated GMM tolerance (rel 1e-5), argmin equal to
// mode 0 except when the two best distances are closer than that.  Mode 0 stays the bit-exact path.
#include "common.h"
#include <cmath>
#include <string>

namespace dsr {

struct GmmModel {      // must mirror k_gmm.hip
  int K, D, G, maxRef;
  std::vector<int> refN, off;
  std::vector<float> mean, ivar, det, val, scale, pi, count;
  std::vector<std::string> cbNames, dsNames;
  DevBuf<int> d_off; DevBuf<float> d_mean, d_ivar; DevBuf<float> d_cst; DevBuf<float> d_val, d_scale; int Dp;
  bool mfmaReady; int KP, GT;
  DevBuf<float> d_A; DevBuf<float> d_bn; DevBuf<int> d_tileCb;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int FT = 256;      // frames per workgroup: two 32-frame column tiles per wavefront, one scanning thread per frame
static constexpr int SC = 64;       // staged codebook scores per frame: flushed at a chunk end once >= 32 are waiting
static constexpr int TS = 36;       // padded row count of the transposed distance tile [frame][row]

// exact distance in the reference's operation order (codebookBasic.cc:509-531); used only to break near ties
__device__ __noinline__ float exact_dist(const float* __restrict__ xr, const float* __restrict__ mu, const float* __restrict__ iv, float cst, int D)
{
  float d = cst;
  for (int i = 0; i < D; i++) { const float df = __fsub_rn(mu[i], xr[i]); d = __fadd_rn(d, __fmul_rn(__fmul_rn(df, df), iv[i])); }
  return d;
}

// closes a codebook for one frame: near ties between the two best candidates are settled in the reference's own arithmetic
__device__ __noinline__ void gmm_finish(const float* __restrict__ x, long nme, long N, int D, int Dp, const float* __restrict__ mean,
                                        const float* __restrict__ ivar, const float* __restrict__ cst, const float* __restrict__ val,
                                        float sl, float m1, int a1, float m2, int a2, int cbStart, float* sdst, unsigned char* adst)
{
  float best = m1; int ba = a1;
  if (m2 - m1 <= 1e-4f * (fabsf(m1) + 1.0f) && nme < N) {
    const float* xr = x + nme * D;
    const float e1 = exact_dist(xr, mean + (size_t) (cbStart + a1) * Dp, ivar + (size_t) (cbStart + a1) * Dp, cst[cbStart + a1], D);
    const float e2 = exact_dist(xr, mean + (size_t) (cbStart + a2) * Dp, ivar + (size_t) (cbStart + a2) * Dp, cst[cbStart + a2], D);
    if (e2 < e1 || (e2 == e1 && a2 < a1)) { best = e2; ba = a2; } else { best = e1; ba = a1; }
  }
  float sc = 0.5f * (best + 2.0f * val[cbStart + ba]);
  if (sl != 1.0f) sc *= sl;
  *sdst = sc; *adst = (unsigned char) ba;
}

// LDS: [Abuf: 2 x S2 x 64 float][tileT: FT x TS float][sbuf: FT x SC float][abuf: FT x SC u8]
template <int S2>   // S2 = KP/2 MFMA steps
__global__ __launch_bounds__(256) void k_gmm_mfma(const float* __restrict__ x, long N, int D, int Dp, int K, int G, int nChunks,
                                                  const int* __restrict__ off, const float* __restrict__ Apack,
                                                  const float* __restrict__ mean, const float* __restrict__ ivar, const float* __restrict__ cst,
                                                  const float* __restrict__ val, const float* __restrict__ scale,
                                                  float* __restrict__ score, unsigned char* __restrict__ argmin)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Abuf = reinterpret_cast<float*>(smem);                 // [2][S2][64] double buffered
  float* tileT = Abuf + 2 * S2 * 64;                             // [FT][TS]
  float* sbuf = tileT + FT * TS;                                 // [FT][SC]
  unsigned char* abuf = reinterpret_cast<unsigned char*>(sbuf + FT * SC);   // [FT][SC]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long n0 = (long) blockIdx.x * FT;
  const int col = lane & 31, kh = lane >> 5;

  // B fragments of this wave's 2 x 32 frames: b[t][s] = B[k = 2s+kh][col]
  float b[2][S2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const long n = n0 + 64 * wave + 32 * t + col;
    const bool live = n < N;
#pragma unroll
    for (int s = 0; s < S2; s++) {
      const int k = 2 * s + kh; float v = 0.0f;
      if (live) {
        if (k < D) { const float q = x[n * D + k]; v = q * q; }
        else if (k < 2 * D) v = x[n * D + (k - D)];
        else if (k == 2 * D) v = 1.0f;
      }
      b[t][s] = v;
    }
  }
  // scan state: thread tid walks the Gaussians of frame n0+tid in order (all threads in lockstep)
  int curK = 0, cbStart = 0, curEnd = __builtin_amdgcn_readfirstlane(off[1]); float m1 = 1E20f, m2 = 1E20f; int a1 = 0, a2 = 0; int kFlush0 = 0;
  const long nme = n0 + tid;

  constexpr int PRE = (S2 * 64 + 255) / 256;
  for (int i = tid; i < S2 * 64; i += 256) Abuf[i] = Apack[i];
  __syncthreads();
  for (int ch = 0; ch < nChunks; ch++) {
    const float* Acur = Abuf + (ch & 1) * S2 * 64; float* Anext = Abuf + ((ch + 1) & 1) * S2 * 64;
    float apre[PRE];                                             // next chunk's operand image, in flight during the MFMAs
    if (ch + 1 < nChunks) {
#pragma unroll
      for (int q = 0; q < PRE; q++) { const int i = tid + 256 * q; apre[q] = (i < S2 * 64) ? Apack[(size_t) (ch + 1) * S2 * 64 + i] : 0.0f; }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
#pragma unroll
    for (int s = 0; s < S2; s++) {
      const float av = Acur[s * 64 + lane];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[0][s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[1][s], acc1, 0, 0, 0);
    }
    // transposed tile: rows (4 q + 8 g' ...) of one frame are contiguous -> 16-byte LDS stores
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row0 = 8 * q + 4 * kh;
      *reinterpret_cast<float4*>(tileT + (64 * wave + col) * TS + row0) = make_float4(acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]);
      *reinterpret_cast<float4*>(tileT + (64 * wave + 32 + col) * TS + row0) = make_float4(acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]);
    }
    if (ch + 1 < nChunks) {
#pragma unroll
      for (int q = 0; q < PRE; q++) { const int i = tid + 256 * q; if (i < S2 * 64) Anext[i] = apre[q]; }
    }
    __syncthreads();
    const int g0 = ch * 32; const int nrow = ((g0 + 32 < G) ? 32 : G - g0);
    // All threads walk the same rows, so codebook boundaries are workgroup-uniform: they live in scalar registers
    // (uniform branches, no exec-mask juggling); the per-row update of the two best distances is branch-free.
#pragma unroll 1
    for (int q = 0; q < 8; q++) {
      if (4 * q >= nrow) break;
      const float4 v4 = *reinterpret_cast<const float4*>(tileT + tid * TS + 4 * q);
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = 4 * q + j;
        if (r < nrow) {
          const float v = vv[j]; const int idx = g0 + r - cbStart;
          const bool lt1 = v < m1, lt2 = v < m2;
          m2 = lt1 ? m1 : (lt2 ? v : m2); a2 = lt1 ? a1 : (lt2 ? idx : a2);
          m1 = lt1 ? v : m1; a1 = lt1 ? idx : a1;
          if (g0 + r + 1 == curEnd) {                            // codebook complete (uniform)
            gmm_finish(x, nme, N, D, Dp, mean, ivar, cst, val, scale[curK], m1, a1, m2, a2, cbStart,
                       sbuf + tid * SC + (curK - kFlush0), abuf + tid * SC + (curK - kFlush0));
            curK++; m1 = 1E20f; m2 = 1E20f; a1 = 0; a2 = 0; cbStart = curEnd;
            curEnd = (curK < K) ? __builtin_amdgcn_readfirstlane(off[curK + 1]) : 0x7FFFFFFF;
          }
        }
      }
    }
    // every thread walks the same rows, so the staged count is uniform: flush contiguous runs once 32 are waiting
    const int cnt = curK - kFlush0;
    if (cnt >= 32 || (curK == K && cnt > 0)) {
      __syncthreads();
      for (int i = tid; i < FT * cnt; i += 256) {
        const int f = i / cnt, c = i - f * cnt; const long n = n0 + f;
        if (n < N) { score[n * K + kFlush0 + c] = sbuf[f * SC + c]; if (argmin) argmin[n * K + kFlush0 + c] = abuf[f * SC + c]; }
      }
      kFlush0 = curK;
    }
    __syncthreads();
  }
}

void gmm_prepare_mfma(GmmModel& m)
{
  if (m.mfmaReady) return;
  if (m.D > 64) throw Error(DSR_E_DIMENSION, "MFMA scoring supports dimN <= 64 (got %d); use mode 0", m.D);
  { const int need = (2 * m.D + 1 + 1) / 2; const int sup[6] = {14, 20, 33, 40, 48, 65}; int S = 65; for (int i = 5; i >= 0; i--) if (sup[i] >= need) S = sup[i]; m.KP = 2 * S; }
  const int S2 = m.KP / 2;
  m.GT = (m.G + 31) / 32;
  std::vector<float> A((size_t) m.GT * S2 * 64, 0.0f);
  for (int g = 0; g < m.G; g++) {
    // codebook of g
    int k = (int) (std::upper_bound(m.off.begin(), m.off.end(), g) - m.off.begin()) - 1;
    double c = (double) (float) (m.pi[k] + m.det[g]);             // float _pi + float det (codebookBasic.cc:481)
    std::vector<float> row(m.KP, 0.0f);
    for (int d = 0; d < m.D; d++) {
      const double iv = m.ivar[(size_t) g * m.D + d], mu = m.mean[(size_t) g * m.D + d];
      row[d] = (float) iv; row[m.D + d] = (float) (-2.0 * mu * iv); c += mu * mu * iv;
    }
    row[2 * m.D] = (float) c;
    const int ch = g / 32, i = g % 32;
    for (int kk = 0; kk < m.KP; kk++) { const int s = kk / 2, kh = kk & 1; A[((size_t) ch * S2 + s) * 64 + kh * 32 + i] = row[kk]; }
  }
  // padding rows of the last chunk must never win
  for (int g = m.G; g < m.GT * 32; g++) { const int ch = g / 32, i = g % 32; const int kk = 2 * m.D; A[((size_t) ch * S2 + kk / 2) * 64 + (kk & 1) * 32 + i] = 1E30f; }
  m.d_A.upload(A);
  m.mfmaReady = true;
}

void gmm_score_mfma(GmmModel& m, const float* x, long N, float* score, unsigned char* argmin, hipStream_t st)
{
  gmm_prepare_mfma(m);
  const int S2 = m.KP / 2;
  const size_t lds = sizeof(float) * ((size_t) 2 * S2 * 64 + (size_t) FT * TS + (size_t) FT * SC) + (size_t) FT * SC;
  dim3 grid(cdiv(N, FT));
#define LAUNCH(SS) { DSR_HIP(hipFuncSetAttribute((const void*) k_gmm_mfma<SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
  hipLaunchKernelGGL(k_gmm_mfma<SS>, grid, dim3(256), lds, st, x, N, m.D, m.Dp, m.K, m.G, m.GT, m.d_off.p, m.d_A.p, m.d_mean.p, m.d_ivar.p, m.d_cst.p, m.d_val.p, m.d_scale.p, score, argmin); }
  switch (S2) {
    case 14: LAUNCH(14) break; case 20: LAUNCH(20) break; case 33: LAUNCH(33) break;
    case 40: LAUNCH(40) break; case 48: LAUNCH(48) break; default: LAUNCH(65) break;
  }
#undef LAUNCH
  DSR_HIP(hipGetLastError());
}

}  // namespace dsr
