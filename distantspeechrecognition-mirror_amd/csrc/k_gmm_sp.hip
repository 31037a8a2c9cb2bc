// csrc/k_gmm_sp.hip -- mode 2 of dsr_gmm_score for codebooks of four Gaussians: the frame x Gaussian contraction of k_gmm_mfma.hip
// (same operands, same trust radius, same tie list -- read its header first) in a software-pipelined shape.
//
// k_gmm_mfma_reg runs two waves per SIMD and hopes that one wave's candidate search falls under the other's MFMAs; measured, the two
// drift into step and a third of the search is paid on top of the contraction (8.25 ms against 5.7 ms of MFMA time at 1 M frames x 4096
// Gaussians).  Here ONE wave per SIMD owns four 32-frame column tiles (128 frames: the Gaussian operand is fetched once for twice the
// frames) and two accumulator sets; the vector work of chunk c - 1 -- sixteen candidate searches, the tie hand-over, the strip's way to
// memory -- is cut into slices that sit BETWEEN the MFMA groups of chunk c, a scheduling barrier after every group pinning them there.
// The slices are straight-line code, so that the scheduler can put VALU work behind every single MFMA:
//   * near ties only set a bit in a per-lane mask; the list is served once per chunk (one branch),
//   * the four -log w of a codebook come as one 16-byte LDS read issued before the comparison chain, the winner's is selected,
//   * the argmin of a strip (32 codebooks: four chunks) is collected in a register, two bits a codebook,
//   * scores and argmins leave through buffer stores whose range check stands in for the "n < N, codebook < K" branches.
// Reference: CodebookBasic::_scoreOpt (asr/gaussian/codebookBasic.cc:509-535).
#include "common.h"
#include "gmm_model.h"
#include <type_traits>

namespace dsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS3 __attribute__((address_space(3)))

template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

// a full tie list: the codebook is settled where it stands, in the reference's operation order (cold: a call keeps it out of the loop's instruction stream)
__device__ __noinline__ unsigned long long gmm_sp_settle(const float* __restrict__ xr, const float* __restrict__ mu, const float* __restrict__ iv, const float* __restrict__ cst, int D, int Dp)
{
  float best = 0.0f; int ba = 0;
  for (int r = 0; r < 4; r++) {
    float d = cst[r];
    for (int i = 0; i < D; i++) { const float df = __fsub_rn(mu[(size_t) r * Dp + i], xr[i]); d = __fadd_rn(d, __fmul_rn(__fmul_rn(df, df), iv[(size_t) r * Dp + i])); }
    if (r == 0 || d < best) { best = d; ba = r; }
  }
  return ((unsigned long long) __float_as_uint(best) << 32) | (unsigned) ba;
}

// DBG (measurement only, DSR_GMM_SPDBG): bit 0 no searches, bit 1 no flush, bit 2 no tie hand-over
template <int S4, int DBG = 0>   // S4 = KP/8: contraction steps in groups of four
__global__ __launch_bounds__(256, 1) void k_gmm_mfma_sp(const float* __restrict__ x, long N, int D, int Dp, int K, int G, int nChunks,
                                                        const float* __restrict__ Apack, const float* __restrict__ mean, const float* __restrict__ ivar,
                                                        const float* __restrict__ cst, const float* __restrict__ val,
                                                        float* __restrict__ score, unsigned char* __restrict__ argmin,
                                                        unsigned long long* __restrict__ tieList, unsigned* __restrict__ tieCount, unsigned tieCap,
                                                        float ivMax2, float termMax)
{
  constexpr int S2 = 4 * S4;                                     // MFMA groups per chunk (one group: the four tiles' MFMAs of one contraction step)
  constexpr int NT = 4, FTW = 32 * NT, SCP = 33;                 // tiles and frames of a wave; pitch of a strip row (32 staged codebooks + 1)
  constexpr int NSRCH = 4 * NT, NFL = 16, FIT = FTW / 2 / NFL;   // search slices; flush slices and the frames-pairs of one
  constexpr int NS = NSRCH + 1 + NFL;                            // slices of a chunk's vector work: searches, tie hand-over, flush
  constexpr unsigned INV = 0x7F000000u;                          // a buffer offset beyond every range: the store is dropped
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform to the compiler too: the wave's frame range and its buffer descriptors stay in SGPRs)
  const int G4 = (G + 3) & ~3;
  float* valL = reinterpret_cast<float*>(smem);                  // [G] -log w of every Gaussian
  LDS3 float* sb3 = (LDS3 float*) (valL + G4 + wave * (FTW * SCP));                     // this wave's strip [FTW frames][SCP]
  LDS3 unsigned* am3 = (LDS3 unsigned*) (valL + G4 + 4 * FTW * SCP) + wave * (FTW * 2);   // its argmin words [FTW][2 (kh)]
  const LDS3 f32x4* val4 = (const LDS3 f32x4*) valL;
  __shared__ unsigned s_tie;
  if (tid == 0) s_tie = 0u;
  for (int i = tid; i < G4; i += 256) valL[i] = i < G ? val[i] : 0.0f;
  __syncthreads();
  const long n0 = (long) blockIdx.x * (4 * FTW) + FTW * wave;    // first frame of this wave
  unsigned long long* myList = tieList + (size_t) blockIdx.x * tieCap;

  float b[NT][S2]; unsigned kLive[NT];                           // kLive: K for a live frame, 0 beyond N ("codebook < K of a live frame" is one compare)
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const long n = n0 + 32 * t + col;
    const bool live = n < N; kLive[t] = live ? (unsigned) K : 0u;
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
  float thrS[NT];                                                // 1e-5 S per frame (k_gmm_mfma.hip header)
#pragma unroll
  for (int t = 0; t < NT; t++) {
    float xx = 0.0f;
#pragma unroll
    for (int s = 0; s < S2; s++) if (2 * s + kh < D) xx += b[t][s];
    xx += __shfl_xor(xx, 32, 64);
    thrS[t] = 1e-5f * (ivMax2 * xx + termMax);
  }
  // the wave's rows of the two outputs as buffers: what lies beyond its live frames (or beyond K: offset INV) is dropped by the range check
  const long nfr = N - n0; const int frames = nfr <= 0 ? 0 : (nfr < FTW ? (int) nfr : FTW);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*) (score + (frames ? n0 * K : 0)), 0, frames * K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*) (argmin ? argmin + (frames ? n0 * K : 0) : (unsigned char*) score), 0, argmin ? frames * K : 0, 0x00020000);
  const unsigned shCol = 2u * (4u * (col >> 3) + ((col >> 1) & 3u));   // where strip column `col` sits in its lane's argmin word
  const int rowBase = col * SCP + kh, amBase = 2 * kh + (col & 1);

  unsigned am[NT] = {0u, 0u, 0u, 0u}; unsigned tieMask = 0u; f32x4 vPre = {0.0f, 0.0f, 0.0f, 0.0f};
  bool flOn = false; unsigned sOff = INV, aOff = INV;
  // ---- the vector work of a finished chunk `pch` (accumulators `prev`), slice by slice; every index below is a compile-time constant
  auto search = [&](const f32x16 (&prev)[NT], const int pch, auto KK) __attribute__((always_inline)) {
    constexpr int k = decltype(KK)::value, t = k % NT, q = k / NT;
    const int kcb = pch * 8 + 2 * q + kh;                        // codebook of accumulator registers 4q .. 4q+3
    const f32x4 v4 = vPre;                                       // the codebook's four -log w: read a slice ago
    float m1 = prev[t][4 * q], m2 = 1E20f; unsigned a1 = 0u; bool w[4];
#pragma unroll
    for (int j = 1; j < 4; j++) {
      const float v = prev[t][4 * q + j];
      const bool lt1 = v < m1, lt2 = v < m2;
      m2 = lt1 ? m1 : (lt2 ? v : m2);
      m1 = lt1 ? v : m1; a1 = lt1 ? (unsigned) j : a1; w[j] = lt1;
    }
    const bool tf = (m2 - m1 <= fmaxf(1e-4f * (fabsf(m1) + 1.0f), thrS[t]) || thrS[t] > 1e-3f * fabsf(m1)) && (unsigned) kcb < kLive[t];
    tieMask |= tf ? (1u << k) : 0u;
    float v0 = v4.x, v1 = v4.y, v2 = v4.z, v3 = v4.w;
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));                 // (as plain registers: selects over LOADED values are turned into branches)
    const float vv = w[3] ? v3 : (w[2] ? v2 : (w[1] ? v1 : v0));               // the winner's -log w, by the chain's own flags (a1: the last j that won)
    const int c4 = pch & 3;                                      // the chunk's place in its strip of four
    sb3[rowBase + 32 * t * SCP + c4 * 8 + 2 * q] = 0.5f * (m1 + 2.0f * vv);
    const unsigned sh = 8u * c4 + 2u * q;
    am[t] = (am[t] & ~(3u << sh)) | (a1 << sh);
    asm volatile("" : "+v"(am[t]), "+v"(tieMask));               // (done here, in this slice: left alone the bookkeeping sinks to the tie slice, sixteen searches' flags alive in SGPRs)
    // the next search's -log w (the next chunk's first after the last; chunk -1 and a phantom last chunk read beside the table: nothing of them is kept)
    vPre = val4[k + 1 < NSRCH ? pch * 8 + 2 * ((k + 1) / NT) + kh : (pch + 1) * 8 + kh];
  };
  auto ties = [&](const int pch) __attribute__((always_inline)) {
    const int c4 = pch & 3;
    while (tieMask) {                                            // rare: about one (frame, codebook) in a thousand
      const int k = __ffs(tieMask) - 1; tieMask &= tieMask - 1u;
      const int t = k % NT, q = k / NT; const int kcb = pch * 8 + 2 * q + kh; const long nme = n0 + 32 * t + col;
      const unsigned slot = atomicAdd(&s_tie, 1u);
      if (slot < tieCap) myList[slot] = ((unsigned long long) nme << 32) | (unsigned long long) (unsigned) kcb;
      else {                                                     // list full: settle it here
        const int cb = kcb * 4;
        const unsigned long long r = gmm_sp_settle(x + nme * D, mean + (size_t) cb * Dp, ivar + (size_t) cb * Dp, cst + cb, D, Dp);
        const unsigned ba = (unsigned) r;
        sb3[rowBase + 32 * t * SCP + c4 * 8 + 2 * q] = 0.5f * (__uint_as_float((unsigned) (r >> 32)) + 2.0f * valL[cb + ba]);
        const unsigned sh = 8u * c4 + 2u * q;
#pragma unroll
        for (int tt = 0; tt < NT; tt++) if (tt == t) am[tt] = (am[tt] & ~(3u << sh)) | (ba << sh);
      }
    }
    // does the strip leave?  (32 codebooks staged, or the last chunk)
    flOn = (unsigned) pch < (unsigned) nChunks && (c4 == 3 || pch + 1 == nChunks);
    if (flOn) {
#pragma unroll
      for (int t = 0; t < NT; t++) am3[(32 * t + col) * 2 + kh] = am[t];
      const int kF = (pch & ~3) * 8; const int cnt = K - kF < 32 ? K - kF : 32;
      sOff = col < cnt ? (unsigned) (kF + col + kh * K) * 4u : INV; aOff = col < cnt ? (unsigned) (kF + col + kh * K) : INV;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
  };
  auto flush_slice = [&](auto SL) __attribute__((always_inline)) {   // 2 x FIT frames of the strip: 128-byte runs of scores, 32-byte runs of argmins
    constexpr int sl = decltype(SL)::value;
    if (flOn) {
      int Kl = K; asm volatile("" : "+s"(Kl));                   // (opaque: or the 2 x 64 per-lane offsets of a strip are hoisted out of the chunk loop and spilled)
#pragma unroll
      for (int i = 0; i < FIT; i++) {
        const int f2 = 2 * (sl * FIT + i), f = kh + f2;          // (the whole offset in the VGPR: the range check does not see an SGPR offset)
        const float v = sb3[f * SCP + col]; const unsigned w = am3[amBase + 2 * f2];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, sOff + (unsigned) (f2 * Kl) * 4u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char) ((w >> shCol) & 3u), ra, aOff + (unsigned) (f2 * Kl), 0, 0);
      }
      if constexpr (sl == NFL - 1) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }
  };
  auto vslice = [&](const f32x16 (&prev)[NT], const int pch, auto KK) __attribute__((always_inline)) {
    constexpr int k = decltype(KK)::value;
    if constexpr (k < NSRCH) { if constexpr (!(DBG & 1)) search(prev, pch, KK); else if constexpr (k == 0) { if (prev[0][0] + prev[1][1] + prev[2][2] + prev[3][3] == 123.456f) sb3[lane] = 1.0f; } }
    else if constexpr (k == NSRCH) { if constexpr (!(DBG & 4)) ties(pch); }
    else { if constexpr (!(DBG & 2)) flush_slice(std::integral_constant<int, k - NSRCH - 1>{}); }
  };
  // ---- chunk ch: its MFMA groups, the previous chunk's slices between them.  The Gaussian operand is replaced in place: entry s4 is
  // re-loaded for the next chunk right after its last MFMA group, a whole chunk ahead of its use.
  const f32x4* Ap4 = reinterpret_cast<const f32x4*>(Apack) + lane;
  f32x4 aop[S4];
#pragma unroll
  for (int q = 0; q < S4; q++) aop[q] = Ap4[q * 64];
  auto contract = [&](f32x16 (&acc)[NT], const f32x16 (&prev)[NT], const int ch) __attribute__((always_inline)) {
    const f32x4* nextA = Ap4 + (size_t) ((ch + 1 < nChunks) ? ch + 1 : nChunks - 1) * S4 * 64;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
    static_for<0, S2>([&](auto II) __attribute__((always_inline)) {
      constexpr int idx = decltype(II)::value, s4 = idx >> 2, j = idx & 3;
#pragma unroll
      for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[s4][j], b[t][idx], acc[t], 0, 0, 0);
      static_for<(idx * NS) / S2, ((idx + 1) * NS) / S2>([&](auto KK) __attribute__((always_inline)) { vslice(prev, ch - 1, KK); });
      if constexpr (j == 3) aop[s4] = nextA[s4 * 64];
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // chunks in pairs (set A, set B).  Before the first there is "chunk -1" (zero accumulators; its codebooks are < 0: no tie, no flush, its
  // strip columns are overwritten); an odd count ends with a phantom chunk (the last operand again) whose codebooks lie beyond K.
  f32x16 accA[NT], accB[NT];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) accB[t][i] = 0.0f;
  const int nChunksP = (nChunks + 1) & ~1;
  for (int ch = 0; ch < nChunksP; ch += 2) { contract(accA, accB, ch); contract(accB, accA, ch + 1); }
  static_for<0, NS>([&](auto KK) __attribute__((always_inline)) { vslice(accB, nChunksP - 1, KK); });
  __syncthreads();
  if (tid == 0) tieCount[blockIdx.x] = s_tie;
}

// frames per workgroup of the shape above (the tie list is segmented by workgroup: gmm_score_mfma sizes it with this)
int gmm_sp_frames() { return 512; }

size_t gmm_sp_lds(const GmmModel& m) { return sizeof(float) * ((size_t) ((m.G + 3) & ~3) + (size_t) 4 * 128 * 33) + (size_t) 4 * 128 * 2 * sizeof(unsigned); }

// launches the scoring kernel (the caller runs k_gmm_ties over the list afterwards); false when the model's shape has no instantiation
bool gmm_sp_launch(GmmModel& m, const float* x, long N, float* score, unsigned char* argmin, unsigned long long* tieList, unsigned* tieCount, unsigned cap, hipStream_t st)
{
  const int S4 = m.KP / 8; const size_t lds = gmm_sp_lds(m);
  if (lds > 160 * 1024 - 64) return false;
  dim3 grid((unsigned) cdiv(N, (long) gmm_sp_frames()));
#define LS(SS) { DSR_HIP(hipFuncSetAttribute((const void*) k_gmm_mfma_sp<SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
  hipLaunchKernelGGL((k_gmm_mfma_sp<SS>), grid, dim3(256), lds, st, x, N, m.D, m.Dp, m.K, m.G, m.GT, m.d_bn.p, m.d_mean.p, m.d_ivar.p, m.d_cst.p, m.d_val.p, \
                     score, argmin, tieList, tieCount, cap, 2.0f * m.ivMax, m.termMax); }
  if (S4 == 10 && getenv("DSR_GMM_SPDBG")) {
    const int dbg = atoi(getenv("DSR_GMM_SPDBG"));
#define LD(DD) { DSR_HIP(hipFuncSetAttribute((const void*) k_gmm_mfma_sp<10, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
  hipLaunchKernelGGL((k_gmm_mfma_sp<10, DD>), grid, dim3(256), lds, st, x, N, m.D, m.Dp, m.K, m.G, m.GT, m.d_bn.p, m.d_mean.p, m.d_ivar.p, m.d_cst.p, m.d_val.p, \
                     score, argmin, tieList, tieCount, cap, 2.0f * m.ivMax, m.termMax); }
    switch (dbg) { case 1: LD(1) break; case 2: LD(2) break; case 3: LD(3) break; case 4: LD(4) break; case 6: LD(6) break; case 7: LD(7) break; default: LS(10) break; }
#undef LD
    DSR_HIP(hipGetLastError());
    return true;
  }
  switch (S4) { case 4: LS(4) break; case 5: LS(5) break; case 9: LS(9) break; case 10: LS(10) break; case 12: LS(12) break; case 17: LS(17) break; default: return false; }
#undef LS
  DSR_HIP(hipGetLastError());
  return true;
}

}  // namespace dsr
