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
// Numerics: fp32 fmaf chain in k order (exact-f32 MFMA).  The expanded form is off the reference's (mu-x)^2*iv accumulation by at most
// (n + 2) 2^-24 S per distance, n = 2 dimN + 1 terms, S = sum of the terms' magnitudes <= 2 ivMax |x|^2 + termMax (gmm_model.h: model-wide
// maxima, |x|^2 per frame) -- the bound grows with the CANCELLED terms, not with the distance (means far from zero: S >> distance).  Every
// (frame, codebook) whose two best candidates lie within 1e-5 S (>= twice that bound) of each other, or whose bound is no longer small against
// the distance itself (1e-5 S > 1e-3 |d|), is re-scored in the reference's own arithmetic over ALL Gaussians of the codebook (k_gmm_ties):
// argmin = mode 0's on every frame; re-scored entries carry mode 0's score bits, the others agree with mode 0 to rel 2e-6 on
// well-conditioned models (asserted in tests/test_gpu_parity.py) and to 1e-3 by construction.  Mode 0 stays the bit-exact path.
#include "common.h"
#include "gmm_model.h"
#include <algorithm>
#include <cmath>
#include <string>

namespace dsr {


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

// closes a codebook for one frame; when the expanded form cannot be trusted (see the header) the whole codebook is re-scored in the reference's arithmetic
__device__ __noinline__ void gmm_finish(const float* __restrict__ x, long nme, long N, int D, int Dp, const float* __restrict__ mean,
                                        const float* __restrict__ ivar, const float* __restrict__ cst, const float* __restrict__ val,
                                        float sl, float m1, int a1, float m2, float thrS, int cbStart, int cbN, float* sdst, unsigned char* adst)
{
  float best = m1; int ba = a1;
  if ((m2 - m1 <= fmaxf(1e-4f * (fabsf(m1) + 1.0f), thrS) || thrS > 1e-3f * fabsf(m1)) && nme < N) {
    const float* xr = x + nme * D; best = 0.0f; ba = 0;
    for (int r = 0; r < cbN; r++) {
      const float e = exact_dist(xr, mean + (size_t) (cbStart + r) * Dp, ivar + (size_t) (cbStart + r) * Dp, cst[cbStart + r], D);
      if (r == 0 || e < best) { best = e; ba = r; }
    }
  }
  float sc = 0.5f * (best + 2.0f * val[cbStart + ba]);
  if (sl != 1.0f) sc *= sl;
  *sdst = sc; *adst = (unsigned char) ba;
}

template <int S2>   // S2 = KP/2 MFMA steps
__global__ __launch_bounds__(256) void k_gmm_mfma(const float* __restrict__ x, long N, int D, int Dp, int K, int G, int nChunks,
                                                  const int* __restrict__ off, const float* __restrict__ Apack,
                                                  const float* __restrict__ mean, const float* __restrict__ ivar, const float* __restrict__ cst,
                                                  const float* __restrict__ val, const float* __restrict__ scale,
                                                  float* __restrict__ score, unsigned char* __restrict__ argmin, float ivMax2, float termMax)
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
  // |x|^2 of the scanning thread's own frame -> 1e-5 S, the trust radius of the expanded form (header)
  float thrS;
  { float xx = 0.0f; const long nq = n0 + tid; if (nq < N) for (int d = 0; d < D; d++) { const float q = x[nq * D + d]; xx += q * q; } thrS = 1e-5f * (ivMax2 * xx + termMax); }
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
            gmm_finish(x, nme, N, D, Dp, mean, ivar, cst, val, scale[curK], m1, a1, m2, thrS, cbStart, curEnd - cbStart,
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Candidate search in the accumulator layout (codebooks of R = 4, 8, 16 or 32 Gaussians, all the same size).
// v_mfma_f32_32x32x2_f32 leaves, in lane l, column (frame) l % 32 and the sixteen rows (Gaussians) 8 (i / 4) + 4 (l / 32) + i % 4, i = 0..15:
//   R = 4 : the four registers acc[4q .. 4q+3] ARE one codebook (number 2q + l/32 of the chunk's eight): minimum, second minimum and their
//           indices are found with a dozen vector instructions per codebook, no LDS tile, no cross-lane traffic;
//   R >= 8: a codebook's rows are split evenly between the lane and its partner l ^ 32; each finds the two best of its R/2 registers and one
//           exchange merges them -- ties go to the smaller Gaussian index, as the reference's strict '<' in file order.
// Near ties (the two best closer than 1e-4 relative: the expanded form cannot order them) are NOT settled inside this kernel: a call -- or
// even an inlined 2 x 39-step exact loop -- in the middle of the contraction costs more than the contraction (80 live operand registers
// around every call: measured 24 ms of 33).  The lane appends (frame, codebook, the two candidates) to a list in memory and moves on;
// k_gmm_ties settles the list afterwards in the reference's own arithmetic, one thread per entry, and overwrites score and argmin.  A list that
// is full sends the lane through the exact computation on the spot (cold code, never reached at the list size the launcher picks).
// DSR_GMM_DBG (measurement only): bit 0 contraction alone, bit 1 no tie test, bit 2 no score stores, bit 3 no codebook close.  1 M frames x 1024 x 4:
// 9.2 ms whole, 5.3 contraction alone, 7.9 without the tie test, 8.0 without the stores, 6.7 without both, 6.1 without close and stores.  Tried on top and
// dropped (tools/ab_gmm.sh): near ties staged in LDS (+-0: the cost of the tie branch is its divergence, not its store), the strip flushed at the top of the
// next chunk behind a pinned operand prefetch with 16-byte stores (+8 %).
// Scores wait in an LDS strip [frame][64 codebooks] (pitch 65: conflict free) and leave as 128-byte runs.  One wave owns two 32-frame tiles (B
// fragments in registers); the Gaussian operand is read from LDS four contraction steps at a time (one ds_read_b128 per four MFMA pairs).
__device__ __forceinline__ float exact_dist_inl(const float* __restrict__ xr, const float* __restrict__ mu, const float* __restrict__ iv, float cst, int D)
{
  float d = cst;
  for (int i = 0; i < D; i++) { const float df = __fsub_rn(mu[i], xr[i]); d = __fadd_rn(d, __fmul_rn(__fmul_rn(df, df), iv[i])); }
  return d;
}
// (the list is cut into one segment per workgroup of the contraction kernel, filled through a counter in that workgroup's LDS: a single
// counter in memory serialises a million atomics at one L2 channel -- measured 8 ms)
// Sixteen lanes settle one entry: the rows of the two candidates (mean, inverse variance) and the frame come in as contiguous 16-lane loads (a thread
// per entry walked five arrays at a stride of a row each: 300 M scattered 4-byte requests per million frames, 0.9 ms), every lane forms the terms
// ((mu - x)^2 iv) of its dimensions, and the sums run over the terms in the reference's order d = 0, 1, ... -- each term handed round the group by a
// lane exchange, every lane of the group adding the same numbers.
__global__ __launch_bounds__(256) void k_gmm_ties(const unsigned long long* __restrict__ list, const unsigned* __restrict__ counts, unsigned cap, const float* __restrict__ x, int D, int Dp,
                                                  int K, int R, const float* __restrict__ mean, const float* __restrict__ ivar, const float* __restrict__ cst, const float* __restrict__ val,
                                                  const float* __restrict__ scale, float* __restrict__ score, unsigned char* __restrict__ argmin)
{
  unsigned cnt = counts[blockIdx.x]; if (cnt > cap) cnt = cap;
  const int lane = threadIdx.x & 63, l16 = lane & 15, gbase = lane & 48;
  const unsigned grp = threadIdx.x >> 4, ngrp = blockDim.x >> 4;
  constexpr int NJ = 4;                                          // dimensions per lane: D <= 64
  for (unsigned i0 = 0; i0 < cnt; i0 += ngrp) {                  // (uniform trip count: the exchanges below need every lane of the wave)
    const unsigned i = i0 + grp; const bool live = i < cnt;
    const unsigned long long e = live ? list[(size_t) blockIdx.x * cap + i] : 0ull;
    const long n = (long) (e >> 32); const int k = (int) (e & 0xFFFFFFFFull);
    const int cb = k * R; const float* xr = x + n * D;
    float xv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) { const int d = l16 + 16 * j; xv[j] = (live && d < D) ? xr[d] : 0.0f; }
    float best = 0.0f; int ba = 0;
    for (int r0 = 0; r0 < R; r0 += 2) {                          // two Gaussians at a time (R is even): their loads and sums side by side
      const float* m1 = mean + (size_t) (cb + r0) * Dp; const float* v1 = ivar + (size_t) (cb + r0) * Dp;
      const float* m2 = m1 + Dp; const float* v2 = v1 + Dp;
      float t1[NJ], t2[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int d = l16 + 16 * j; t1[j] = 0.0f; t2[j] = 0.0f;
        if (live && d < D) {
          const float d1 = __fsub_rn(m1[d], xv[j]), d2 = __fsub_rn(m2[d], xv[j]);
          t1[j] = __fmul_rn(__fmul_rn(d1, d1), v1[d]); t2[j] = __fmul_rn(__fmul_rn(d2, d2), v2[d]);
        }
      }
      float e1 = live ? cst[cb + r0] : 0.0f, e2 = live ? cst[cb + r0 + 1] : 0.0f;
#pragma unroll
      for (int j = 0; j < NJ; j++)
        for (int q = 0; q < 16; q++) {
          if (16 * j + q >= D) break;                            // (uniform)
          e1 = __fadd_rn(e1, __shfl(t1[j], gbase + q, 64)); e2 = __fadd_rn(e2, __shfl(t2[j], gbase + q, 64));
        }
      if (r0 == 0 || e1 < best) { best = e1; ba = r0; }          // strict '<' in file order: the first of equals wins (codebookBasic.cc:524-531)
      if (e2 < best) { best = e2; ba = r0 + 1; }
    }
    if (live && l16 == 0) {
      float sc = 0.5f * (best + 2.0f * val[cb + ba]);
      const float sl = scale[k]; if (sl != 1.0f) sc *= sl;
      score[n * K + k] = sc; if (argmin) argmin[n * K + k] = (unsigned char) ba;
    }
  }
}

// Every wavefront runs on its own: 64 frames (two MFMA column tiles, B fragments in registers), the Gaussian operand straight from memory
// (it is 1.3 MB and lives in L2; one fully coalesced 16-byte load per lane feeds four MFMA pairs; the next chunk's loads are in flight during
// this chunk's MFMAs), a wave-private LDS strip for the scores.  No workgroup barrier inside the contraction.
template <int S4, int R>   // S4 = KP/8: contraction steps in groups of four
__global__ __launch_bounds__(256, 2) void k_gmm_mfma_reg(const float* __restrict__ x, long N, int D, int Dp, int K, int G, int nChunks,
                                                      const float* __restrict__ Apack, const float* __restrict__ mean, const float* __restrict__ ivar,
                                                      const float* __restrict__ cst, const float* __restrict__ val, const float* __restrict__ scale, int unitScale,
                                                      float* __restrict__ score, unsigned char* __restrict__ argmin,
                                                      unsigned long long* __restrict__ tieList, unsigned* __restrict__ tieCount, unsigned tieCap, int valInLds, int dbg,
                                                      float ivMax2, float termMax)
{
  constexpr int S2 = 4 * S4;
  constexpr int CPC = 32 / R;                                    // codebooks per 32-row chunk
  constexpr int SCP = 33;                                        // pitch of a strip row: 32 staged codebooks + 1
  constexpr int FLUSH = 32 / CPC;                                // chunks between flushes: 32 codebooks staged
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* valL = reinterpret_cast<float*>(smem);                  // [G] -log w of every Gaussian (when it fits)
  float* sbuf = valL + (valInLds ? ((G + 3) & ~3) : 0) + wave * (64 * SCP);      // this wave's [64 frames][SCP]
  unsigned char* abuf = reinterpret_cast<unsigned char*>(valL + (valInLds ? ((G + 3) & ~3) : 0) + 4 * 64 * SCP) + wave * (64 * 32);   // [64][32]
  const long n0 = (long) blockIdx.x * FT + 64 * wave;            // first frame of this wave
  const int col = lane & 31, kh = lane >> 5;
  __shared__ unsigned s_tie;
  if (tid == 0) s_tie = 0u;
  if (valInLds) for (int i = tid; i < G; i += 256) valL[i] = val[i];
  __syncthreads();
  unsigned long long* myList = tieList + (size_t) blockIdx.x * tieCap;

  float b[2][S2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const long n = n0 + 32 * t + col;
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
  // 1e-5 S per frame of the two tiles (header): |x|^2 is the sum of the operand's squared half, the lane's steps + its partner's
  float thrS[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    float xx = 0.0f;
#pragma unroll
    for (int s = 0; s < S2; s++) if (2 * s + kh < D) xx += b[t][s];
    xx += __shfl_xor(xx, 32, 64);
    thrS[t] = 1e-5f * (ivMax2 * xx + termMax);
  }
  const float4* Ap4 = reinterpret_cast<const float4*>(Apack) + lane;
  int kFlush0 = 0;                                               // first codebook of the strip
  // closes one codebook for one frame: score from the best candidate; a near tie goes to the list
  auto finish = [&](const long nme, const int fr, const int kcb, const float m1, const int a1, const float m2, const float thr) __attribute__((always_inline)) {
    float best = m1; int ba = a1;
    if (!(dbg & 2) && (m2 - m1 <= fmaxf(1e-4f * (fabsf(m1) + 1.0f), thr) || thr > 1e-3f * fabsf(m1)) && nme < N) {
      const unsigned slot = atomicAdd(&s_tie, 1u);
      if (slot < tieCap) myList[slot] = ((unsigned long long) nme << 32) | (unsigned long long) (unsigned) kcb;
      else {                                                     // list full: settle it here (cold)
        const int cb = kcb * R; const float* xr = x + nme * D;
        for (int r = 0; r < R; r++) {
          const float e = exact_dist_inl(xr, mean + (size_t) (cb + r) * Dp, ivar + (size_t) (cb + r) * Dp, cst[cb + r], D);
          if (r == 0 || e < best) { best = e; ba = r; }
        }
      }
    }
    float vv;                                                    // (two branches: "valInLds ? valL[i] : val[i]" through one pointer is a FLAT load, which pays
    if (valInLds) { vv = ((const __attribute__((address_space(3))) float*) valL)[kcb * R + ba]; asm volatile("" ::: "memory"); }   // the memory path's latency even in LDS)
    else vv = val[kcb * R + ba];
    float sc = 0.5f * (best + 2.0f * vv);
    if (!unitScale) { const float sl = scale[kcb]; if (sl != 1.0f) sc *= sl; }
    sbuf[fr * SCP + (kcb - kFlush0)] = sc; abuf[fr * 32 + (kcb - kFlush0)] = (unsigned char) ba;
  };
  // search of one codebook group (tile t, group q) of a finished chunk
  auto search = [&](const f32x16 (&acc)[2], const int ch, const int t, const int q) __attribute__((always_inline)) {
    const long nme = n0 + 32 * t + col; const int fr = 32 * t + col;
    if (R == 4) {
      const int kcb = ch * 8 + 2 * q + kh;                       // codebook of registers 4q .. 4q+3
      float m1 = acc[t][4 * q], m2 = 1E20f; int a1 = 0, a2 = 0;
#pragma unroll
      for (int j = 1; j < 4; j++) {
        const float v = acc[t][4 * q + j];
        const bool lt1 = v < m1, lt2 = v < m2;
        m2 = lt1 ? m1 : (lt2 ? v : m2); a2 = lt1 ? a1 : (lt2 ? j : a2);
        m1 = lt1 ? v : m1; a1 = lt1 ? j : a1;
      }
      if (dbg & 8) { if (m1 + m2 == 123.456f) sbuf[fr] = (float) (a1 + a2); }
      else if (kcb < K) finish(nme, fr, kcb, m1, a1, m2, thrS[t]);
    } else {
      constexpr int RH = R / 2;                                  // registers of a codebook in this lane
      const int c = q;                                           // q counts the chunk's codebooks here
      const int kcb = ch * CPC + c;
      float m1 = 1E20f, m2 = 1E20f; int a1 = 0, a2 = 0;
#pragma unroll
      for (int i = 0; i < RH; i++) {
        const float v = acc[t][c * RH + i]; const int idx = 8 * (i >> 2) + 4 * kh + (i & 3);          // row inside the codebook
        const bool lt1 = v < m1, lt2 = v < m2;
        m2 = lt1 ? m1 : (lt2 ? v : m2); a2 = lt1 ? a1 : (lt2 ? idx : a2);
        m1 = lt1 ? v : m1; a1 = lt1 ? idx : a1;
      }
      // merge with the partner lane's two best: order by (value, index)
      const float p1 = __shfl_xor(m1, 32, 64), p2 = __shfl_xor(m2, 32, 64); const int pa = __shfl_xor(a1 | (a2 << 8), 32, 64);
      const int q1 = pa & 255, q2 = pa >> 8;
      auto before = [](float va, int ia, float vb, int ib) { return va < vb || (va == vb && ia < ib); };
      float r1, r2; int s1;                                      // (the runner-up's value is all the trust test needs: its index no longer travels)
      if (before(m1, a1, p1, q1)) { r1 = m1; s1 = a1; r2 = before(m2, a2, p1, q1) ? m2 : p1; }
      else { r1 = p1; s1 = q1; r2 = before(p2, q2, m1, a1) ? p2 : m1; }
      if (kh == (c & 1) && kcb < K) finish(nme, fr, kcb, r1, s1, r2, thrS[t]);          // one of the pair closes the codebook
    }
  };
  constexpr int NSRCH = 2 * (R == 4 ? 4 : CPC);                  // search groups per chunk (two tiles)
  auto flush = [&](const int ch) __attribute__((always_inline)) {     // after chunk ch has been searched
    if (((ch + 1) % FLUSH) == 0 || ch + 1 == nChunks) {          // the wave's own strip: 64 frames x <= 32 codebooks leave as 128-byte runs
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      const int staged = (ch + 1) * CPC - kFlush0;
      const int cnt = (kFlush0 + staged < K ? staged : K - kFlush0);
      if (!(dbg & 4)) for (int f = kh; f < 64; f += 2) {
        const long n = n0 + f;
        if (n < N && col < cnt) { score[n * K + kFlush0 + col] = sbuf[f * SCP + col]; if (argmin) argmin[n * K + kFlush0 + col] = abuf[f * 32 + col]; }
      }
      kFlush0 += staged;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
  };
  // Per chunk: the contraction (the next chunk's operand loads in flight), then the search of its accumulators.  Two waves share a SIMD: one's
  // search runs under the other's MFMAs.  (Tried and dropped: searching chunk ch - 1 between the MFMAs of chunk ch inside one wave -- the second
  // accumulator set and the longer live ranges cost the second wave per SIMD, 10.5 -> 11.6 ms.)
  float4 acur[S4];
#pragma unroll
  for (int q = 0; q < S4; q++) acur[q] = Ap4[q * 64];
  for (int ch = 0; ch < nChunks; ch++) {
    float4 anext[S4];
    {                                                            // unconditional (the last chunk re-reads itself): with the loads under a branch the
      const int cn = (ch + 1 < nChunks) ? ch + 1 : ch;           // compiler waits for them at the merge point, before the first MFMA (-3 %)
#pragma unroll
      for (int q = 0; q < S4; q++) anext[q] = Ap4[((size_t) cn * S4 + q) * 64];
    }
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; i++) { acc[0][i] = 0.0f; acc[1][i] = 0.0f; }
#pragma unroll
    for (int s4 = 0; s4 < S4; s4++) {
      const float av[4] = {acur[s4].x, acur[s4].y, acur[s4].z, acur[s4].w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[0][4 * s4 + j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[1][4 * s4 + j], acc[1], 0, 0, 0);
      }
    }
    if (dbg & 1) { if (acc[0][0] + acc[1][3] == 123.456f) sbuf[lane] = 1.0f; }
    else {
#pragma unroll
      for (int g = 0; g < NSRCH; g++) search(acc, ch, g & 1, g >> 1);
      flush(ch);
    }
#pragma unroll
    for (int q = 0; q < S4; q++) acur[q] = anext[q];
  }
  __syncthreads();
  if (tid == 0) tieCount[blockIdx.x] = s_tie;
}

void gmm_prepare_mfma(GmmModel& m)
{
  if (m.mfmaReady) return;
  if (m.D > 64) throw Error(DSR_E_DIMENSION, "MFMA scoring supports dimN <= 64 (got %d); use mode 0", m.D);
  { const int need = (2 * m.D + 1 + 1) / 2; const int sup[6] = {16, 20, 36, 40, 48, 68}; int S = 68; for (int i = 5; i >= 0; i--) if (sup[i] >= need) S = sup[i]; m.KP = 2 * S; }   // multiples of four
  const int S2 = m.KP / 2;
  m.GT = (m.G + 31) / 32;
  std::vector<float> A((size_t) m.GT * S2 * 64, 0.0f);
  double ivMax = 0.0, termMax = 0.0;                              // model-wide maxima behind the rounding bound of the expanded form (header)
  for (int g = 0; g < m.G; g++) {
    // codebook of g
    int k = (int) (std::upper_bound(m.off.begin(), m.off.end(), g) - m.off.begin()) - 1;
    double c = (double) (float) (m.pi[k] + m.det[g]);             // float _pi + float det (codebookBasic.cc:481)
    std::vector<float> row(m.KP, 0.0f);
    for (int d = 0; d < m.D; d++) {
      const double iv = m.ivar[(size_t) g * m.D + d], mu = m.mean[(size_t) g * m.D + d];
      row[d] = (float) iv; row[m.D + d] = (float) (-2.0 * mu * iv); c += mu * mu * iv;
      if (std::fabs(iv) > ivMax) ivMax = std::fabs(iv);
    }
    { const double c0 = (double) (float) (m.pi[k] + m.det[g]); double q = 0.0; for (int d = 0; d < m.D; d++) { const double iv = m.ivar[(size_t) g * m.D + d], mu = m.mean[(size_t) g * m.D + d]; q += std::fabs(mu * mu * iv); }
      const double tm = 2.0 * q + std::fabs(c0); if (tm > termMax) termMax = tm; }
    row[2 * m.D] = (float) c;
    const int ch = g / 32, i = g % 32;
    for (int kk = 0; kk < m.KP; kk++) { const int s = kk / 2, kh = kk & 1; A[((size_t) ch * S2 + s) * 64 + kh * 32 + i] = row[kk]; }
  }
  // padding rows of the last chunk must never win
  for (int g = m.G; g < m.GT * 32; g++) { const int ch = g / 32, i = g % 32; const int kk = 2 * m.D; A[((size_t) ch * S2 + kk / 2) * 64 + (kk & 1) * 32 + i] = 1E30f; }
  m.d_A.upload(A);
  m.ivMax = (float) (ivMax * 1.0000002); m.termMax = (float) (termMax * 1.0000002);      // (rounded up)
  {
    // the same operand with four consecutive steps of a lane side by side: A4[(chunk S4 + s4) 64 + lane][j] = A[(chunk S2 + 4 s4 + j) 64 + lane]
    const int S4 = S2 / 4; std::vector<float> A4(A.size());
    for (int chk = 0; chk < m.GT; chk++) for (int s4 = 0; s4 < S4; s4++) for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++)
      A4[(((size_t) chk * S4 + s4) * 64 + l) * 4 + j] = A[((size_t) chk * S2 + 4 * s4 + j) * 64 + l];
    m.d_bn.upload(A4);
  }
  m.mfmaReady = true;
}

void gmm_score_mfma(GmmModel& m, const float* x, long N, float* score, unsigned char* argmin, hipStream_t st)
{
  gmm_prepare_mfma(m);
  const int S2 = m.KP / 2;
  // codebooks of one size R in {4, 8, 16, 32}: the candidate search stays in the accumulator registers
  int R = m.refN.empty() ? 0 : m.refN[0];
  for (int k = 1; k < m.K; k++) if (m.refN[k] != R) R = 0;
  if ((R == 4 || R == 8 || R == 16 || R == 32) && !getenv("DSR_GMM_MFMA_SCAN") && m.K < 65536 && N < ((long) 1 << 32)) {
    const int S4 = S2 / 4;
    const int valInLds = (m.G <= 8192) ? 1 : 0;
    const size_t ldsR = sizeof(float) * ((size_t) (valInLds ? ((m.G + 3) & ~3) : 0) + (size_t) 4 * 64 * 33) + (size_t) 4 * 64 * 32;
    int unitScale = 1; for (int k = 0; k < m.K; k++) if (m.scale[k] != 1.0f) unitScale = 0;
    // near ties: one list entry each (frame, codebook, two candidates); sized for 1 in 32 (measured: 1 in a few hundred), and a full list
    // is settled in place
    GmmTieScratch& ts = m.tie.at(st);                           // this stream's own (two pipes may score with one model on two streams)
    DevBuf<unsigned long long>& tieList = ts.list; DevBuf<unsigned>& tieCount = ts.count;
    // unit scales, the -log w table in LDS, a shape k_gmm_sp.hip is instantiated for: one wave per SIMD, MFMAs back to back, the short search
    const bool sp = gmm_sp_has(S4, R) && unitScale && gmm_sp_lds(m) <= (size_t) 160 * 1024 - 64 && !(getenv("DSR_GMM_SP") && atoi(getenv("DSR_GMM_SP")) == 0);
    const int FTG = sp ? gmm_sp_frames() : FT;                    // frames per workgroup (the tie list is segmented by workgroup)
    const unsigned nBlk = (unsigned) cdiv(N, FTG);
    // per workgroup (256 frames x K codebooks): room for 1 near tie in 32 (measured: 1 in a thousand); a full segment is settled in place
    unsigned cap = (unsigned) std::min<size_t>(std::max<size_t>((size_t) FTG * (size_t) m.K / 32, 256), ((size_t) 1 << 30) / nBlk);
    tieList.reserve((size_t) nBlk * cap); tieCount.reserve(nBlk);
    if (getenv("DSR_GMM_TIECAP")) cap = std::min<unsigned>(cap, (unsigned) atoi(getenv("DSR_GMM_TIECAP")));   // (tests: a list that fills up, entries settled in place)
    dim3 gridR(nBlk);
    const int dbg = getenv("DSR_GMM_DBG") ? atoi(getenv("DSR_GMM_DBG")) : 0;
#define LR(SS, RR) { DSR_HIP(hipFuncSetAttribute((const void*) k_gmm_mfma_reg<SS, RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsR)); \
  hipLaunchKernelGGL((k_gmm_mfma_reg<SS, RR>), gridR, dim3(256), ldsR, st, x, N, m.D, m.Dp, m.K, m.G, m.GT, m.d_bn.p, m.d_mean.p, m.d_ivar.p, m.d_cst.p, m.d_val.p, m.d_scale.p, unitScale, \
                     score, argmin, tieList.p, tieCount.p, cap, valInLds, dbg, 2.0f * m.ivMax, m.termMax); }
#define LRS(RR) switch (S4) { case 4: LR(4, RR) break; case 5: LR(5, RR) break; case 9: LR(9, RR) break; case 10: LR(10, RR) break; case 12: LR(12, RR) break; default: LR(17, RR) break; }
    if (sp && gmm_sp_launch(m, R, x, N, score, argmin, ts.masks, tieList.p, tieCount.p, cap, st)) { }
    else if (R == 4) LRS(4) else if (R == 8) LRS(8) else if (R == 16) LRS(16) else LRS(32)
#undef LRS
#undef LR
    DSR_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_gmm_ties, dim3(nBlk), dim3(256), 0, st, tieList.p, tieCount.p, cap, x, m.D, m.Dp, m.K, R, m.d_mean.p, m.d_ivar.p, m.d_cst.p, m.d_val.p, m.d_scale.p, score, argmin);
    DSR_HIP(hipGetLastError());
    if (getenv("DSR_GMM_TIES")) {
      std::vector<unsigned> c(nBlk); DSR_HIP(hipStreamSynchronize(st)); DSR_HIP(hipMemcpy(c.data(), tieCount.p, 4 * (size_t) nBlk, hipMemcpyDeviceToHost));
      size_t tot = 0; unsigned mx = 0; for (unsigned v : c) { tot += v; if (v > mx) mx = v; }
      fprintf(stderr, "[dsr gmm] near ties: %zu of %zu (fullest segment %u of %u)\n", tot, (size_t) N * m.K, mx, cap);
    }
    return;
  }
  const size_t lds = sizeof(float) * ((size_t) 2 * S2 * 64 + (size_t) FT * TS + (size_t) FT * SC) + (size_t) FT * SC;
  dim3 grid(cdiv(N, FT));
#define LAUNCH(SS) { DSR_HIP(hipFuncSetAttribute((const void*) k_gmm_mfma<SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
  hipLaunchKernelGGL(k_gmm_mfma<SS>, grid, dim3(256), lds, st, x, N, m.D, m.Dp, m.K, m.G, m.GT, m.d_off.p, m.d_A.p, m.d_mean.p, m.d_ivar.p, m.d_cst.p, m.d_val.p, m.d_scale.p, score, argmin, 2.0f * m.ivMax, m.termMax); }
  switch (S2) {
    case 16: LAUNCH(16) break; case 20: LAUNCH(20) break; case 36: LAUNCH(36) break;
    case 40: LAUNCH(40) break; case 48: LAUNCH(48) break; default: LAUNCH(68) break;
  }
#undef LAUNCH
  DSR_HIP(hipGetLastError());
}

}  // namespace dsr
