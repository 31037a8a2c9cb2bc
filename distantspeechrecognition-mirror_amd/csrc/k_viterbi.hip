// csrc/k_viterbi.hip -- time-synchronous Viterbi token passing over a static WFST, batched over
// utterances, bit-exact with the reference's sequential decoder.
//
// Replaces _Decoder::decode and everything under it (asr/decoder/decoder.h:488-737, 956-1015),
// the _TokenList hash/list (decoder.h:48-320), _Token (asr/lattice/lattice.h:37-79) and
// bestHypo (decoder.h:748-773) for DecoderFlyWeight (decoder.h:1127-1139).
//
// What "bit-exact" needs and how it is kept in a parallel kernel:
//   * token scores live as two floats; every placement is computed in double from them and rounded on
//     construction; recombination compares the UNROUNDED double candidate with the incumbent's
//     float sum (decoder.h:519-528).  That fold is order dependent, so it is replayed literally:
//     every placement of a frame gets its arrival rank ("slot") = its position in the reference's
//     nested loops (token list order x arc order x depth-first epsilon recursion); the placements
//     of one destination state are folded in slot order by one thread.
//   * the next frame's list order is "last inserted first" (decoder.h:246-247, replacement keeps the
//     position): a state's position is decided by its FIRST arrival, so the new list is the
//     first-arrival candidates in reverse slot order -- a stream compaction, no sort.
//   * the beam uses the previous frame's best UNROUNDED emitting placement (decoder.h:521,566-588).
//   * epsilon arcs are expanded depth first without recombination (decoder.h:979-983); the graph's
//     expansion tables (wfst_graph.h) enumerate those paths in the reference's order, and the float
//     rounding of every intermediate epsilon token is reproduced when the path is walked.
// One workgroup (1024 threads) owns one utterance at a time and loops over its frames; workgroups pull
// utterances from a queue.  All arithmetic that decides a comparison is done with explicit
// non-contracted IEEE operations (this file is compiled with -ffp-contract=off).
#include "common.h"
#include "wfst_graph.h"
#include "lattice.h"
#include "lexicon.h"
#include "wordtrace.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace dsr {

struct Tok { int32_t node; float ac; float lm; uint32_t bp; };        // register view; node bit31: edge input == silenceX
// token lists in memory: what a frame's beam test and expansion need (TokA) apart from what only the end phase needs (TokB)
struct TokA { float ac; float lm; uint32_t bp; uint32_t xs; };        // xs: first expansion record of the node | bit31: edge input == silenceX
struct TokB { int32_t node; int32_t cnt; };                           // cnt: number of expansion records of the node
// expansion record as the register path reads it: what the expansion needs in the first 16 bytes, what the new token needs in the second
// eps1: cost of the first epsilon hop (meta bit17: it has an output).  Further hops (meta bits 18..30: hop h = 1..13 has an output):
// a two-hop path carries the second cost in p2 (float bits), longer ones the offset of their hop costs in GraphDev::pathCost;
// meta bit31: more than 14 hops, walked through the path/arc arrays instead.
struct XRecD { int32_t dist; float cost; uint32_t meta; float eps1; int32_t dst; int32_t p2; int32_t dstXoff; int32_t dstCnt; };
struct Side { double ttl; float ac; float lm; int32_t rec; uint32_t prevBp; int32_t c; uint32_t next; };   // a later arrival at an occupied state
struct CandA { double ttl; float ac; float lm; };
struct CandB { int32_t dst; int32_t next; int32_t rec; uint32_t prevBp; };   // rec bit30: the emitting arc's input is the silence symbol
struct Bp { uint32_t prev; uint32_t rec; };

static constexpr uint32_t kNone = 0xFFFFFFFFu;
static constexpr uint32_t kEndBit = 0x80000000u;
static constexpr int kThreads = 1024;             // 16 waves per CU at 128 VGPRs (measured in round 2: 512 x 256 VGPRs 22 % slower, 768 x 168 VGPRs 4 % slower; two 512-thread workgroups per CU 1.2x slower)
static constexpr int kWaves = kThreads / 64;
static constexpr int kFastK = 8;
static constexpr int kB = 2;                        // placements whose loads are in flight together (batch of the register-path phases)                  // placements a thread keeps in registers on the register path
static constexpr int kFastC = 24576;               // most placements per frame on the register path (those beyond kFastK per thread are parked in memory)
static constexpr int kFastE = 8190;                // most expanding tokens per frame on the register path
static constexpr int kP1 = 16;                     // token rounds per wave in the register path's beam pass
static constexpr int kW = 4;                        // register placements whose P6 loads are in flight together (8: 3 % slower; parked ones: kB)
static constexpr int kSideLds = 496;               // later arrivals kept in LDS (the rest go to memory)

struct GraphDev {
  int nNodes, initial;
  const int* xoff; const XRec* xrec; const XRecD* xrecD; const int* xarc; const int* xpathOff;
  const int* eoff; const ERec* erec; const int* path; const float* pathCost;
  const float* arcCost; const uint32_t* arcOut; const uint32_t* arcIn;
  const int* nodeFinal; const float* nodeCost;
};

// Time slicing (segFrames > 0): a work item is one SEGMENT of an utterance -- segFrames frames -- and the items are taken in the order segment-major, utterance-minor,
// so all utterances of a batch advance together and end together.  (Run to completion, a workgroup per utterance, the workgroups end over a span of one utterance's
// duration once the queue is empty: 11 % of the launch at 1000 utterances on 256 CUs.)  Between its segments an utterance is its token list + these scalars.
struct SegState { int n, status, maxActive, pad; long arenaOff, chunkEnd, arenaUsed; double thresh; long long stat[3]; };

struct DecDev {
  double beam, lmScale, lmPenalty, silPenalty; uint32_t silenceX; int noPen;
  // time slicing: frames per segment (0: off), segments per utterance, queues (8: one per XCD, 1: one for all), the pool of back-pointer records and how many of
  // them an utterance takes at a time (a barrier pair and a device atomic each time: 9 us)
  int segFrames, segCount, segQueues, segDrop; long poolCap, poolChunk;   /* segDrop (tests): bit x set = the workgroups on XCD x do not serve their own queue */ unsigned long long* poolNext; SegState* segState; int* segDone; TokA* saveA; TokB* saveB;
  int maxTok, maxCand; long arenaCap;
  // per-slot scratch (slot s at base + s*stride)
  TokA* tokA; TokB* tokB; TokA* ctok; Side* side; int fastOK; int* tokOff; int* tokCnt; int* owner; int* rank; int* chead; CandA* cA; CandB* cB; unsigned* first; unsigned* tags; Bp* arena;
  int* queue; long long* prof;          // prof: optional per-phase wall-clock ticks (DSR_VITERBI_PROF), 16 per slot
  // dump (slot 0 only)
  int dumpOn; long dumpCap; long* dumpFrameOff; int* dumpNode; float* dumpAc; float* dumpLm; int* dumpArc; long* dumpCount;
  // lattice bookkeeping (generateLattice, decoder.h:531-541,805-953): EVERY placement of every frame is kept, per utterance, in arrival order --
  // {ac, lm, record, parent back pointer} + its unrounded total (the reference's 'worse' chains are an order-dependent function of exactly
  // these; the host replays them, lattice.cpp) -- plus, per back-pointer record, the placement that won its state, and the final token list.
  int latOn; long latCap; uint4* lat; double* latTtl; long* latFrameOff; int* arenaLat; int4* latFinal; int* latInfo;
  // topN > 0 (decoder.h:571-581): a frame expands the topN best tokens of the list in order of their scores and applies no beam; third token buffer
  int topN; TokA* tokA3; TokB* tokB3;
};

// every argument of the kernel, one struct in the kernarg segment (read through KP, see k_viterbi)
struct VitArgs {
  GraphDev G; DecDev D;
  const float* scores; const int* nframesArr; int U, Tmax, nDist;
  dsr_decode_result* res; int* arcsOut; unsigned* wordsOut; int maxPath, useLdsRow, hashN, regionB, cntCap;
};
typedef const __attribute__((address_space(4))) VitArgs* KP;

__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// What an utterance carries from one CU to another (time slicing) is READ with device-scope loads (past the reader's L1).  The utterance never leaves its XCD --
// every XCD has its own queue -- so the L2 both CUs share is the point of coherence and the writes stay plain (the L1 writes through).  Measured alternatives: release /
// acquire fences at device scope write back and invalidate a whole L2 per hand-over (0.1 ms each); write-through stores for the back-pointer records cost 2 % of the launch.
__device__ __forceinline__ void st_u64_dev(void* p, unsigned long long v) { __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64_dev(const void* p) { return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// inclusive prefix sum over the wave: four row shifts inside every 16-lane row, then the row totals are handed on
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- six DPP adds, no LDS round trips
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
  return v;
}
// full-wave min / max by DPP row shifts and row broadcasts (as the scan above): vector-ALU speed, the result in lane 63, handed out as a uniform value.
// (__shfl_xor goes through the LDS crossbar: six dependent ds_bpermute round trips, ~0.3 us per reduction, twice that for a double)
__device__ __forceinline__ float wave_max_f_uni(float v)
{
#define DSR_DPP_F(ctrl, rm) { const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rm, 0xF, false)); v = fmaxf(v, o); }
  DSR_DPP_F(0x111, 0xF) DSR_DPP_F(0x112, 0xF) DSR_DPP_F(0x114, 0xF) DSR_DPP_F(0x118, 0xF) DSR_DPP_F(0x142, 0xA) DSR_DPP_F(0x143, 0xC)
#undef DSR_DPP_F
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_min_d_uni(double v)
{
#define DSR_DPP_D(ctrl, rm) { const int lo = __double2loint(v), hi = __double2hiint(v); \
    const double o = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, ctrl, rm, 0xF, false), __builtin_amdgcn_update_dpp(lo, lo, ctrl, rm, 0xF, false)); v = (o < v) ? o : v; }
  DSR_DPP_D(0x111, 0xF) DSR_DPP_D(0x112, 0xF) DSR_DPP_D(0x114, 0xF) DSR_DPP_D(0x118, 0xF) DSR_DPP_D(0x142, 0xA) DSR_DPP_D(0x143, 0xC)
#undef DSR_DPP_D
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_min_d(double v)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(v, d, 64); v = (o < v) ? o : v; }
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(v, d, 64); v = (o < v) ? o : v; }
  return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
// clears a result record with a zero the optimiser cannot see through: the constant {0, 0, 0} tuple of a memset otherwise lives from the kernel's
// first instruction to its last, in scratch memory, and its 96-bit reload is what trips the register-alignment bug mentioned below
__device__ __forceinline__ void clear_result(dsr_decode_result* r)
{
  int z = 0; asm volatile("" : "+v"(z));
  int* p = reinterpret_cast<int*>(r);
#pragma unroll 1
  for (int i = 0; i < (int) (sizeof(dsr_decode_result) / 4); i++) p[i] = z;
}
// LDS-typed views: "cond ? lds[i] : mem[i]" is compiled as a select of two generic pointers and ONE flat load -- a flat load pays the memory path's
// latency even when it hits LDS.  Reads through these types are ds_read instructions; the two cases are kept in separate branches (fence below).
typedef __attribute__((address_space(3))) float lds_float_t;
typedef __attribute__((address_space(3))) Side lds_side_t;
#define DSR_NO_MERGE() asm volatile("" ::: "memory")
// a whole token as one 16-byte load (three of its four fields used -> the compiler narrows the load to 96 bits: same bug as below)
__device__ __forceinline__ TokA ld_tok(const TokA* p)
{
  const unsigned long long a = reinterpret_cast<const volatile unsigned long long*>(p)[0], b = reinterpret_cast<const volatile unsigned long long*>(p)[1];
  TokA t; t.ac = __uint_as_float((unsigned) (a & 0xFFFFFFFFull)); t.lm = __uint_as_float((unsigned) (a >> 32)); t.bp = (uint32_t) (b & 0xFFFFFFFFull); t.xs = (uint32_t) (b >> 32);
  return t;
}
// a winner's {ac, lm, rec} out of a side record: an 8-byte and a 4-byte load on purpose -- as three adjacent fields the compiler merges them into one
// 96-bit load, and a 96-bit value that gets spilled trips a register-alignment bug of this compiler on gfx950 ("requires even aligned vector registers")
__device__ __forceinline__ void side_winner(const Side* sideL, const Side* side, int sideLds, int idx, float& ac, float& lm, int& rec)
{
  if (idx < sideLds) {
    const lds_side_t* p = (const lds_side_t*) sideL + idx;
    const unsigned long long a = *reinterpret_cast<const volatile __attribute__((address_space(3))) unsigned long long*>(&p->ac);
    rec = *reinterpret_cast<const volatile __attribute__((address_space(3))) int*>(&p->rec);
    ac = __uint_as_float((unsigned) (a & 0xFFFFFFFFull)); lm = __uint_as_float((unsigned) (a >> 32));
    DSR_NO_MERGE();
  } else {
    const Side* p = side + idx;
    const unsigned long long a = *reinterpret_cast<const volatile unsigned long long*>(&p->ac); rec = *reinterpret_cast<const volatile int*>(&p->rec);
    ac = __uint_as_float((unsigned) (a & 0xFFFFFFFFull)); lm = __uint_as_float((unsigned) (a >> 32));
  }
}
// {slot, unrounded total, next} and {next, total} of a side record
__device__ __forceinline__ void side_link(const Side* sideL, const Side* side, int sideLds, int idx, int& c, double& ttl, unsigned& next)
{
  if (idx < sideLds) { const lds_side_t* p = (const lds_side_t*) sideL + idx; c = p->c; ttl = p->ttl; next = p->next; DSR_NO_MERGE(); }
  else { const Side* p = side + idx; c = p->c; ttl = p->ttl; next = p->next; }
}
__device__ __forceinline__ float side_score(const Side* sideL, const Side* side, int sideLds, int idx)
{
  float r;
  if (idx < sideLds) { const lds_side_t* p = (const lds_side_t*) sideL + idx; r = __fadd_rn(p->ac, p->lm); DSR_NO_MERGE(); }
  else { const Side* p = side + idx; r = __fadd_rn(p->ac, p->lm); }
  return r;
}
// "uniform base + 32-bit byte offset": the address form the hardware adds for free (global_load v, v_off, s[base]); a 64-bit element index costs a
// sign extension, a 64-bit shift and a 64-bit add per access.  Callers guarantee index * sizeof < 2^32 (checked on the host: DecoderState::fastOK).
template <class T> __device__ __forceinline__ const T* at32(const void* base, uint32_t byteOff) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byteOff); }
template <class T> __device__ __forceinline__ T* at32w(void* base, uint32_t byteOff) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byteOff); }
// wave-uniform values computed from LDS land in vector registers; these move them to scalar ones
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// One decoded utterance per loop iteration of a persistent workgroup.
//
// A frame runs on one of two paths that produce identical lists:
//   * the register path (frames with at most kFastC placements, 16 per thread): thread t owns the arrival slots t, t+nthr, ...
//     and keeps those placements in registers from expansion to the write of the new list; the expanding
//     tokens are compacted first (slot offsets in LDS), recombination goes through the LDS state table, and only the
//     later arrivals at an occupied state (a few percent) are spilled to memory for the first arrival to fold;
//   * the memory path (any size, and the end expansion): placements are staged in global arrays, one thread per placement.
template <int MODES>
__global__ __launch_bounds__(kThreads) void k_viterbi(const VitArgs argsInKernarg)
{
  // The arguments are read from the kernarg segment where they are used, through a pointer that is made opaque again at every phase boundary
  // (RELOAD): a field is a scalar load from the constant cache in the phase that needs it and dead after it.  Taken by value the two structs are
  // ~130 scalar registers that live from the first instruction to the last; with the per-slot pointers and loop state derived from them the kernel
  // needed 444 more scalars than the 102 a wave has, and every use in the hot phases was a v_readlane reload from a spill lane.
  (void) argsInKernarg;
  constexpr bool PROF = (MODES & 1) != 0, EXTRA = (MODES & 2) != 0;      // EXTRA: lattice bookkeeping, topN, token dump compiled in
  const KP KA = (KP) __builtin_amdgcn_kernarg_segment_ptr();
  KP ka = KA;
#define RELOAD() do { ka = KA; asm volatile("" : "+s"(ka)); } while (0)
  RELOAD();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* srow = reinterpret_cast<float*>(smem);                       // [nDist] when useLdsRow
  // per-frame open-addressing table in LDS: destination state -> first-arrival slot (hashN buckets, 0 = unused).
  // Global atomics execute at the memory side on gfx950 (no L2 residency); the LDS table keeps the recombination
  // traffic on chip.  Frames with more placements than the table can take fall back to the tagged global table.
  const int useLdsRow = ka->useLdsRow, nDist = ka->nDist, hashN = ka->hashN, regionB = ka->regionB, cntCap = ka->cntCap;   // (the LDS layout: live for the whole kernel)
  unsigned* hkey = reinterpret_cast<unsigned*>(srow + (useLdsRow ? ((nDist + 3) & ~3) : 4));
  unsigned* hfirst = hkey + hashN;
  Side* sideL = reinterpret_cast<Side*>(hfirst + hashN);                        // [regionB / 32] later arrivals at an occupied state
  // [cntCap] expansion counts of the list P6 wrote, by list position: the next frame's P1 scans them without waiting for the tokens to come back from memory
  unsigned short* cntL = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(hfirst + hashN) + regionB);
  __shared__ int s_waveTot[kWaves];
  __shared__ int s_waveTotE[kWaves];
  __shared__ double s_waveMin[kWaves];
  __shared__ float s_waveMag[kWaves];
  __shared__ unsigned long long s_waveKey[kWaves];
  __shared__ int s_sideN;
  __shared__ int s_tb[2];            // traceback: hops, words
  __shared__ unsigned s_bm[kFastC / 32];             // register path: slots where an expanding token's run starts
  // register path, per group of 64 slots: expanding tokens that start before the group (13 bits) | how far into its token's run the group's first slot is (19 bits)
  __shared__ unsigned s_gbase[kFastC / 64 + 4];
  __shared__ int s_cnt[(kFastK + 32) * kWaves];
  __shared__ int s_err;             // register path: first arrivals per (k, wave) group, then their exclusive prefix
  __shared__ long long s_prof[32]; __shared__ long long s_tlast;
  // per-utterance statistics and the cold loop state (thread 0 updates them once per frame; they used to ride in scalar registers through every phase)
  __shared__ long long s_stat[3];    // activeHypos, placements, registerFrames
  __shared__ int s_maxActive; __shared__ unsigned s_tag; __shared__ long long s_latOff;
  if (PROF && threadIdx.x < 32) s_prof[threadIdx.x] = 0;
#define TICK(ix) do { if (PROF && tid == 0) { const long long tn = (long long) wall_clock64(); s_prof[ix] += tn - s_tlast; s_tlast = tn; } } while (0)
  constexpr int nthr = kThreads, nw = kWaves;
  __shared__ int s_u, s_seg, s_help; __shared__ long long s_chunk;

  const int tid = threadIdx.x;
  const int slot = blockIdx.x;
  // token lists: buffers 0 and 1 of the slot (current / next, swapped every frame), buffer 2 the spare of topN mode; pointers are formed where they are used
#define TOKA(ix) ((EXTRA && (ix) == 2) ? ka->D.tokA3 + (size_t) slot * ka->D.maxTok : ka->D.tokA + ((size_t) slot * 2 + (size_t) (ix)) * ka->D.maxTok)
#define TOKB(ix) ((EXTRA && (ix) == 2) ? ka->D.tokB3 + (size_t) slot * ka->D.maxTok : ka->D.tokB + ((size_t) slot * 2 + (size_t) (ix)) * ka->D.maxTok)
#define curA TOKA(bufCur)
#define nxtA TOKA(bufNxt)
#define curB TOKB(bufCur)
#define nxtB TOKB(bufNxt)
#define sprA TOKA(bufSpr)
#define sprB TOKB(bufSpr)
#define ctok (ka->D.ctok + (size_t) slot * 8192)
#define side (ka->D.side + (size_t) slot * kFastC)
  // (the staging arrays of the memory path)
#define cA (ka->D.cA + (size_t) slot * ka->D.maxCand)
#define cB (ka->D.cB + (size_t) slot * ka->D.maxCand)
  // first[] holds (tag << 24 | slot); tags count DOWN so every entry of an older frame compares larger and never
  // needs resetting; the table is wiped when the 8-bit tag runs out (and on the very first use of a slot)
  if (tid == 0) s_tag = ka->D.tags[slot];
  // back pointers: one arena per slot (lattice mode: one per utterance -- they outlive the slot: the host builds the lattice from them)
#define arena (ka->D.arena + (size_t) ((EXTRA && ka->D.latOn) ? u : (ka->D.segFrames > 0 ? 0 : slot)) * ka->D.arenaCap)
#define sc (ka->scores + (size_t) u * ka->Tmax * nDist)
  constexpr int fastCapC = ((kFastK + 32) * nthr < kFastC) ? (kFastK + 32) * nthr : kFastC;
  constexpr int fastCapN = kP1 * 64 * nw;                                   // kP1 rounds of 64 tokens per wave
  const bool fastOK = ka->D.fastOK && hashN >= 8192;
  // capacities that follow from the LDS budget of this launch: table (load <= 0.75 per pass), slot offsets, LDS side records
  const int tableC = (hashN >> 1) + (hashN >> 2), eCap = kFastE, sideLds = regionB >> 5;

  // time slicing: the XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0) picks its queue and its share of the utterances (u = xcd mod 8)
  // (segQueues == 1: one queue, any workgroup may take any utterance up -- the hand-over then pays device-scope fences; small grids and the tests)
  const int nq = (EXTRA || ka->D.segFrames <= 0) ? 1 : ka->D.segQueues;
  const int xcd = nq == 8 ? (int) (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : 0;
  // help > 0: this workgroup's own queue is empty and it looks into queue (xcd + help) mod 8 for utterances NOBODY HAS STARTED.  With workgroups on every XCD
  // there are none by then; this is the net under the one assumption the XCD-bound queues make (a queue whose XCD got no workgroup of the grid would otherwise
  // never be served): such an utterance is decoded here in one go -- no hand-over, so no coherence question -- and marked so that its own queue passes it by.
  // (the queue being served lives in LDS, the work item comes out of it as (utterance, segment): nothing of this rides in registers through the frame loop)
  if (tid == 0) s_help = (nq == 8 && ((ka->D.segDrop >> xcd) & 1)) ? 1 : 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      const int segS0 = EXTRA ? 0 : ka->D.segFrames; int help = s_help, uu = -1, sg = 0;
      for (;;) {
        const int qx = (xcd + help) & (nq - 1), Uq = (ka->U - qx + nq - 1) / nq;     // (nq is 1 or 8); utterances of this queue: u = qx mod nq
        int it = -1;
        if (help == 0) { it = atomicAdd(ka->D.queue + qx, 1); if (it >= (segS0 > 0 ? ka->D.segCount * Uq : Uq)) it = -1; }
        else {                                                                // first-segment items only, and only by compare-and-swap: a later item of that queue is not ours to take
          int c = ld_i32(ka->D.queue + qx);
          while (c < Uq) { const int prev = atomicCAS(ka->D.queue + qx, c, c + 1); if (prev == c) { it = c; break; } c = prev; }
        }
        if (it >= 0) { uu = segS0 > 0 ? qx + nq * (it % Uq) : it; sg = segS0 > 0 ? it / Uq : 0; break; }
        if (segS0 > 0 && nq == 8 && help < 7) help++; else break;
      }
      if (uu >= 0 && help > 0) { st_i32(&ka->D.segState[uu].status, -2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_i32(&ka->D.segDone[uu], 0x7FFFFFFF); sg = -1; }   // taken whole
      s_help = help; s_u = uu; s_seg = sg;
    }
    __syncthreads();
    RELOAD();
    const int segS = EXTRA ? 0 : ka->D.segFrames;                           // segS > 0: time slicing (never with the lattice / topN / dump modes)
    const int u = s_u; if (u < 0) break;
    const bool whole = s_seg < 0; const int seg = whole ? 0 : s_seg;        // whole: an utterance of another queue that nobody had started, decoded here in one go
    const int T = ka->nframesArr[u] < ka->Tmax ? ka->nframesArr[u] : ka->Tmax;
    const int fr0 = seg * segS, frEnd = (segS > 0 && !whole) ? fr0 + segS : 0x7FFFFFFF;   // this item: frames fr0 .. frEnd-1 (the end expansion is "frame" T)
    if (fr0 > T) continue;                                                     // the utterance ended in an earlier segment
    if (EXTRA && ka->D.latOn && tid == 0) ka->D.latFrameOff[(size_t) u * (ka->Tmax + 3)] = 0;
    if (PROF && tid == 0) { const long long tn = (long long) wall_clock64(); if (s_prof[15]) s_prof[10] += tn - s_prof[15]; s_tlast = tn; }
#define dump (EXTRA && ka->D.dumpOn && slot == 0)

    int status = DSR_OK;
    if (T <= 0) status = DSR_E_ITERATOR;         // no frame at all: the exception escapes decode() (decoder.h:691)

    int bufCur = 0, bufNxt = 1, bufSpr = 2;
    int n = 1; long arenaOff = 0, chunkEnd = 0, arenaUsed = 0;             // arenaUsed: back-pointer records of the utterance so far (time slicing: its runs of the pool are not contiguous)
    double thresh = HUGE_VAL, topScore = HUGE_VAL;
    for (int i = tid; i < 2 * hashN; i += nthr) hkey[i] = (i < hashN) ? 0u : 0xFFFFFFFFu;
    if (seg == 0) {
      if (tid == 0) {
        s_stat[0] = 0; s_stat[1] = 0; s_stat[2] = 0; s_maxActive = 0; s_latOff = 0;
        TokA t0; t0.ac = 0.0f; t0.lm = 0.0f; t0.bp = kNone; t0.xs = (uint32_t) ka->G.xoff[ka->G.initial]; curA[0] = t0;
        TokB b0; b0.node = ka->G.initial; b0.cnt = ka->G.xoff[ka->G.initial + 1] - ka->G.xoff[ka->G.initial]; curB[0] = b0;
      }
    } else {
      // the segment before may still be running on another CU: wait for it, then take the utterance over (token list into buffer 0, scalars)
      if (tid == 0) { while (ld_i32(&ka->D.segDone[u]) < seg) __builtin_amdgcn_s_sleep(16); }
      __syncthreads();
      if (nq == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const SegState* const sst = ka->D.segState + u;
      status = ld_i32(&sst->status);
      if (status != DSR_OK) continue;                                          // it failed there: its result is written
      n = ld_i32(&sst->n); arenaOff = (long) ld_u64_dev(&sst->arenaOff); chunkEnd = (long) ld_u64_dev(&sst->chunkEnd); arenaUsed = (long) ld_u64_dev(&sst->arenaUsed); thresh = __longlong_as_double((long long) ld_u64_dev(&sst->thresh));
      if (tid == 0) { s_stat[0] = (long long) ld_u64_dev(&sst->stat[0]); s_stat[1] = (long long) ld_u64_dev(&sst->stat[1]); s_stat[2] = (long long) ld_u64_dev(&sst->stat[2]); s_maxActive = ld_i32(&sst->maxActive); s_latOff = 0; }
      const TokA* const svA = ka->D.saveA + (size_t) u * ka->D.maxTok; const TokB* const svB = ka->D.saveB + (size_t) u * ka->D.maxTok;
      for (int i = tid; i < n; i += nthr) {
        const unsigned long long a0 = ld_u64_dev(&svA[i]), a1 = ld_u64_dev(reinterpret_cast<const unsigned long long*>(&svA[i]) + 1), b0 = ld_u64_dev(&svB[i]);
        unsigned long long* da = reinterpret_cast<unsigned long long*>(&curA[i]); da[0] = a0; da[1] = a1; *reinterpret_cast<unsigned long long*>(&curB[i]) = b0;
      }
    }
    __syncthreads();

    bool cntOK = false;                                                        // cntL holds the counts of the current list, and every token of it passes the beam
    int preC = -1;                                                             // >= 0: the frame before has already laid out this frame's slots (bitmap, group table): preC placements
    const bool preRow = useLdsRow && nDist <= 2 * nthr;
    float rowNext[2] = {0.0f, 0.0f};
    if (preRow && fr0 < T) { const float* r0 = sc + (size_t) fr0 * nDist; if (tid < nDist) rowNext[0] = r0[tid]; if (tid + nthr < nDist) rowNext[1] = r0[tid + nthr]; }
    TICK(11);
    // frames 0..T-1 (mode 0), then the end expansion (mode 1) -- with time slicing: this segment's share of them
    int fr = fr0;
    for (; fr <= T && fr < frEnd && status == DSR_OK; fr++) {
      RELOAD();
      // the thread index behind an opaque copy, once per frame: nothing derived from it (lane masks, per-thread addresses of any
      // phase or of the memory path) can be hoisted out of the frame loop, where it would sit in scratch memory and be re-read
      int tidO = (int) threadIdx.x; asm volatile("" : "+v"(tidO));
      const int tid = tidO, lane = tid & 63, wave = tid >> 6;
      const int mode = (fr == T) ? 1 : 0;
      if (mode == 0 && useLdsRow) {
        if (preRow) {
          // the frame's score row was fetched into registers a frame ago (the load -> LDS store round trip was ~0.9 us on every frame's critical path)
          if (tid < nDist) srow[tid] = rowNext[0];
          if (tid + nthr < nDist) srow[tid + nthr] = rowNext[1];
          if (fr + 1 < T) {
            const float* nr = sc + (size_t) (fr + 1) * nDist;
            if (tid < nDist) rowNext[0] = nr[tid];
            if (tid + nthr < nDist) rowNext[1] = nr[tid + nthr];
          }
        } else { for (int i = tid; i < nDist; i += nthr) srow[i] = sc[(size_t) fr * nDist + i]; }
      }
      const float* rowG = sc + (size_t) fr * nDist;                            // the frame's score row in memory
      int numNew = 0, numStat = -1;                                           // tokens written to the new list / tokens the reference's list would hold
      if (fr > 0) TICK(23);                                                   // end of the frame before -> here
      if (PROF && tid == 0) s_tlast = (long long) wall_clock64();
      bool fast = fastOK && mode == 0 && n <= fastCapN && !(EXTRA && (ka->D.latOn || ka->D.topN > 0));   // lattice bookkeeping needs every placement in memory: the memory path has them
      int Cfr = 0;                                                             // placements of this frame (statistics)

      if (fast) {
        // ======================= register path =======================
        const bool pre = preC >= 0;                                            // the slots of this frame were laid out when the list was written (P6): no P1, no P2
        // ---- P1: beam test, per-wave exclusive scans of the placement counts and of the expanding tokens.
        // Token loads are issued eight at a time (straight-line, clamped indices) so their latencies overlap.
        const int chunkT = ((n + nw * 64 - 1) / (nw * 64)) * 64;
        int C = preC, E = n;
        if (!pre) {
        float psc[kP1]; int pcn[kP1]; unsigned pk[kP1];                        // score; expansion count; slot offset | token index << 15 | bit31: expands
        int runC = 0, runE = 0;
        {
          const int b0 = wave * chunkT, b1 = (b0 + chunkT < n) ? b0 + chunkT : n;
#pragma unroll
          for (int g = 0; g < kP1; g += 8) {
            if (g * 64 < chunkT && cntOK) {
              // the list was written by the register path with pruning on: every token is within the beam (it was tested against this very threshold
              // when it was written) and its expansion count waits in LDS
#pragma unroll
              for (int it = g; it < g + 8; it++) { int i = b0 + it * 64 + lane; i = (i < n) ? i : n - 1; pcn[it] = (int) cntL[i]; psc[it] = -HUGE_VALF; }
            } else if (g * 64 < chunkT) {
#pragma unroll
              for (int it = g; it < g + 8; it++) {
                int i = b0 + it * 64 + lane; i = (i < n) ? i : n - 1;
                const float2 a = *reinterpret_cast<const float2*>(&curA[i].ac); pcn[it] = curB[i].cnt;
                psc[it] = __fadd_rn(a.x, a.y);
              }
            } else {
#pragma unroll
              for (int it = g; it < g + 8; it++) { pcn[it] = 0; psc[it] = 0.0f; }
            }
          }
          TICK(16);
          for (int i = tid; i < kFastC / 32; i += nthr) s_bm[i] = 0u;
          TICK(17);
#pragma unroll
          for (int it = 0; it < kP1; it++) {
            const int base = b0 + it * 64; pk[it] = 0u;
            if (base < b1) {
              const int i = base + lane; int cnt = 0;
              if (i < b1) {
                if (!((double) psc[it] > thresh)) cnt = pcn[it];               // beam (decoder.h:586-588)
              }
              pcn[it] = cnt;
              const int incl = wave_incl_scan(cnt, lane);
              const unsigned long long bal = __ballot(cnt > 0);
              if (cnt > 0) pk[it] = 0x80000000u | (unsigned) ((runC + incl - cnt) & 0x7FFF) | ((unsigned) ((runE + __popcll(bal & ((1ull << lane) - 1ull))) & 0xFFFF) << 15);
              runC += __builtin_amdgcn_readlane(incl, 63); runE += __popcll(bal);
            }
          }
          TICK(18);
          if (lane == 0) { s_waveTot[wave] = runC; s_waveTotE[wave] = runE; }
          if (tid == 0) { s_sideN = 0; s_gbase[0] = 0; s_err = 0; }
        }
        __syncthreads();
        TICK(0);
        int cbase, ebase;
        {                                                                      // lane w reads wave w's totals: two scans instead of a serial walk over 32 LDS words
          const int a = (lane < nw) ? s_waveTot[lane] : 0, b = (lane < nw) ? s_waveTotE[lane] : 0;
          const int sa = wave_incl_scan(a, lane), sb = wave_incl_scan(b, lane);
          C = __builtin_amdgcn_readlane(sa, 63); E = __builtin_amdgcn_readlane(sb, 63);
          const int wu = uni(wave);
          cbase = __builtin_amdgcn_readlane(sa - a, wu); ebase = __builtin_amdgcn_readlane(sb - b, wu);
        }
        TICK(19);
        if (!(C > fastCapC || E > eCap || C > 2 * tableC)) {
          // ---- P2: compact list of the expanding tokens (their slot offsets in LDS, the tokens themselves in memory); a bitmap
          // of the slots where a token's run starts and the token count before every group of 64 slots turn "slot -> token"
          // into a population count
          // (when every token expands -- the rule once the lists are pruned at write time -- compact index == list index and the
          // list itself serves as the compact list: no copy)
          RELOAD();
          const bool ident = (E == n);
#pragma unroll
          for (int it = 0; it < kP1; it++) if (pk[it] & 0x80000000u) {
            const int e = ebase + (int) ((pk[it] >> 15) & 0xFFFFu); const int off = cbase + (int) (pk[it] & 0x7FFFu);
            if (!ident) ctok[e] = curA[wave * chunkT + it * 64 + lane];
            atomicOr(&s_bm[off >> 5], 1u << (off & 31));
            for (int g = (off >> 6) + 1; g <= ((off + pcn[it]) >> 6); g++) s_gbase[g] = (unsigned) (e + 1) | ((unsigned) (64 * g - off) << 13);   // this token covers slot 64g-1
          }
          __syncthreads();
        }
        } else __syncthreads();                                                // (laid out by the frame before; this barrier: the score row is in LDS)
        if (C > fastCapC || E > eCap || C > 2 * tableC) { fast = false; __syncthreads(); }     // uniform: the memory path redoes the frame
        else {
          RELOAD();
          const bool ident = (E == n);
          const TokA* __restrict__ ctk = ident ? curA : ctok;
          TICK(1);
          Cfr = C;
          RELOAD();
          // (tq/lq/wq = tid/lane/wave behind an opaque copy: keeps the per-slot address arithmetic of the unrolled phases inside
          // the frame loop -- hoisted out of it, those hundred-odd invariants would live in scratch memory)
          int tq = tid; asm volatile("" : "+v"(tq)); const int lq = tq & 63, wq = tq >> 6;
          // ---- P3: slots c = k * nthr + tid (neighbouring lanes expand neighbouring arcs of the same few tokens).
          // A placement is four words: ac, lm, expansion record (bit30: silence arc) and ek = compact token index | table
          // bucket << 13 (after the fold: bit31 | side index when a later arrival won).  The first kFastK per thread stay in
          // registers for the whole frame; slots beyond (one more batch of eight) are parked in memory between the phases.
          const int K = (C + nthr - 1) / nthr;
          float qac[kFastK], qlm[kFastK]; int qrec[kFastK]; unsigned ek[kFastK];
          double* const ttlS = reinterpret_cast<double*>(cA);                  // unrounded totals, read back by later arrivals only
          uint4* const ovf = reinterpret_cast<uint4*>(cB);                     // parked placements [slot - kFastK * nthr]
          const XRecD* const xrecD = ka->G.xrecD; const float* const pathCost = ka->G.pathCost; const uint32_t silenceX = ka->D.silenceX;
          double locMin = HUGE_VAL; float locMag = 0.0f;                       // locMag: largest |ac| + |lm| of the frame's placements (bounds the rounding of a score, see P4)
          const bool prevNull0 = (fr == 0);                                    // only the start token has no edge (decoder.h:960)
          const double lmS = ka->D.lmScale, lsPen = __dmul_rn(ka->D.lmScale, ka->D.lmPenalty), lsSil = __dmul_rn(ka->D.lmScale, ka->D.silPenalty);
          const bool sil0 = (0u == silenceX);
#pragma unroll
          for (int k = 0; k < kFastK; k++) { qac[k] = 0.0f; qlm[k] = 0.0f; qrec[k] = 0; ek[k] = 0u; }

          const int nPass = (C > tableC) ? 2 : 1;
          auto table_insert = [&](const unsigned dst, const unsigned prod, const int c) __attribute__((always_inline)) -> unsigned {
            const unsigned key = dst + 1u;
            unsigned h = (prod >> 7) & (unsigned) (hashN - 1); int probes = 0;
            for (;;) {
              const unsigned kk = atomicCAS(&hkey[h], 0u, key);
              if (kk == 0u || kk == key) break;
              h = (h + 1u) & (unsigned) (hashN - 1);                           // (triangular steps instead of linear ones: 16.5 vs 16.6 ms, nothing)
              if (++probes > hashN) { s_err = 1; break; }                      // table full: cannot happen below its capacity; fail loudly, never spin
            }
            atomicMin(&hfirst[h], (unsigned) c);
            return h;
          };
          // slot -> (compact token index, position in its expansion list), then the token: score halves into the placement's own registers,
          // record index = first record of the node + position, bit29 of ek = the token's edge was a silence edge
          auto locate = [&](const int k, int& rec, unsigned& ekk) __attribute__((always_inline)) {
            const int c = k * nthr + tq; const int grp = k * nw + wq;
            unsigned e = 0u; int j = 0;
            if (c < C) {
              const unsigned long long W = ((unsigned long long) s_bm[2 * grp + 1] << 32) | s_bm[2 * grp];
              const unsigned gb = s_gbase[grp]; const unsigned long long mine = W & ((2ull << lq) - 1ull);
              e = (gb & 0x1FFFu) + (unsigned) __popcll(mine) - 1u;
              // position in the token's run: slots since the run's start -- inside this group (highest start bit at or below the lane), or carried in from
              // the group before (no second, dependent LDS read of the token's slot offset)
              j = mine ? lq - (63 - __clzll((long long) mine)) : lq + (int) (gb >> 13);
            }
            ekk = e; rec = j;
          };
          auto tokload = [&](float& ac, float& lm, int& rec, unsigned& ekk) __attribute__((always_inline)) {
            const TokA t = *at32<TokA>(ctk, ekk * 16u);
            ac = t.ac; lm = t.lm; rec += (int) (t.xs & 0x7FFFFFFFu); ekk |= (t.xs >> 31) << 29;
          };
          auto expandR = [&](auto NOPEN, const int g8, float* ac8, float* lm8, int* rec8, unsigned* ek8) __attribute__((always_inline)) {
            // NP: both penalty products are zero and no placement's lm can be -0.0 (checked on the host, DecoderState::noPen): "l + 0.0" is then
            // the identity and the four conditional additions of a placement -- an add and two selects each -- are left out
            constexpr bool NP = decltype(NOPEN)::value;
            int4 xr[kB]; int2 xd[kB]; bool tsil[kB];
#pragma unroll
            for (int i = 0; i < kB; i++) { tsil[i] = (ek8[i] >> 29) & 1u; ek8[i] &= 0x1FFFu; }
#pragma unroll
            for (int i = 0; i < kB; i++) {                                      // eight record loads in flight
              xr[i] = *at32<int4>(xrecD, (uint32_t) rec8[i] * 32u); xd[i] = *at32<int2>(xrecD, (uint32_t) rec8[i] * 32u + 16u);
            }
            if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TICK(28); }
#pragma unroll
            for (int i = 0; i < kB; i++) {
              const int c = (g8 + i) * nthr + tq;
              if (c < C) {
                const int xdist = xr[i].x; const float xcost = __int_as_float(xr[i].y); const uint32_t xmeta = (uint32_t) xr[i].z;
                const float e1 = __int_as_float(xr[i].w);
                const int plen = (int) (xmeta & 0xFFFFu);
                // selects in place of branches: the additions that may not apply are computed and dropped (same values, one
                // basic block -- the scalar operands are fetched once per placement instead of once per branch)
                const double lm0 = (double) lm8[i];
                const bool has = plen != 0;
                double lmNode;
                {                                                              // first epsilon hop (decoder.h:979-983): its cost travels with the record
                  double l = __dadd_rn(lm0, __dmul_rn(lmS, (double) e1));
                  if (!NP) {
                    const double l1 = __dadd_rn(l, lsPen); l = (xmeta & 0x20000u) ? l1 : l;
                    const double l2 = __dadd_rn(l, lsSil); l = (sil0 && (prevNull0 || !tsil[i])) ? l2 : l;   // prevIn != silenceX <=> the token's edge was no silence edge
                  }
                  lmNode = has ? (double) (float) l : lm0;
                }
                const bool pnull = has ? false : prevNull0;
                const bool pinSil = has ? sil0 : tsil[i];                      // after an epsilon hop the edge input is 0
                if (plen > 1) {
                  if (!(xmeta & 0x80000000u)) {
                    float hc0 = 0.0f, hc1 = 0.0f, hc2 = 0.0f;                  // three and four hops (and the head of longer paths): their costs in flight together
                    if (plen > 2) { const float* pc = pathCost + xd[i].y; hc0 = pc[0]; hc1 = pc[1]; hc2 = pc[(plen > 3) ? 2 : 1]; }
                    for (int h = 1; h < plen; h++) {                           // hop costs: inline (two hops) or one independent load per hop
                      const float ch = (plen == 2) ? __int_as_float(xd[i].y) : (h == 1 ? hc0 : h == 2 ? hc1 : h == 3 ? hc2 : pathCost[xd[i].y + h - 1]);
                      double l = __dadd_rn(lmNode, __dmul_rn(lmS, (double) ch));
                      if (!NP && ((xmeta >> (17 + h)) & 1u)) l = __dadd_rn(l, lsPen);
                      lmNode = (double) (float) l;                              // (the edge before is an epsilon edge: no silence penalty possible here)
                    }
                  } else {
                    const int* pp = ka->G.path + ka->G.xpathOff[rec8[i]];
                    for (int h = 1; h < plen; h++) {
                      const int a = pp[h];
                      double l = __dadd_rn(lmNode, __dmul_rn(lmS, (double) ka->G.arcCost[a]));
                      if (!NP && ka->G.arcOut[a] != 0) l = __dadd_rn(l, lsPen);
                      lmNode = (double) (float) l;
                    }
                  }
                }
                double lm = __dadd_rn(lmNode, __dmul_rn(lmS, (double) xcost));
                if (!NP) { const double lm1 = __dadd_rn(lm, lsPen); lm = (xmeta & 0x10000u) ? lm1 : lm; }
                const bool silArc = ((uint32_t) (xdist + 1) == silenceX);
                if (!NP) { const double lm2 = __dadd_rn(lm, lsSil); lm = (silArc && (pnull || !pinSil)) ? lm2 : lm; }
                float rowv;                                                    // (two branches, not "useLdsRow ? srow[..] : rowG[..]": that is one FLAT load)
                if (useLdsRow) { rowv = ((const lds_float_t*) srow)[xdist]; DSR_NO_MERGE(); } else rowv = rowG[xdist];
                const double ac = __dadd_rn((double) ac8[i], (double) rowv);
                const double ttl = __dadd_rn(ac, lm);
                *at32w<double>(ttlS, (uint32_t) c * 8u) = ttl; ac8[i] = (float) ac; lm8[i] = (float) lm; rec8[i] |= (silArc ? 0x40000000 : 0);
                if (ttl < locMin) locMin = ttl;                                // _topScore
                locMag = fmaxf(locMag, __fadd_rn(fabsf(ac8[i]), fabsf(lm8[i])));
                // state table: claim the bucket, keep the smallest slot (with two passes, the other half of the states waits)
                if (PROF) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); TICK(29); }
                const unsigned prod = (unsigned) xd[i].x * 2654435761u;
                if (nPass > 1 && (prod >> 31)) ek8[i] |= 1u << 27;
                else ek8[i] |= (table_insert((unsigned) xd[i].x, prod, c) << 13) | (1u << 28);      // bit28: in the table, not folded yet
                if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TICK(30); }
              }
            }
          };
          auto expand8 = [&](auto NOPEN, const int g8, float* ac8, float* lm8, int* rec8, unsigned* ek8) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < kB; i++) locate(g8 + i, rec8[i], ek8[i]);
            if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TICK(26); }
#pragma unroll
            for (int i = 0; i < kB; i++) tokload(ac8[i], lm8[i], rec8[i], ek8[i]);
            if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TICK(27); }
            expandR(NOPEN, g8, ac8, lm8, rec8, ek8);
          };
          auto park_store = [&](const int kb, const float* oac, const float* olm, const int* orec, const unsigned* oek) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < kB; i++) { const int c = (kb + i) * nthr + tq; if (c < C) ovf[c - kFastK * nthr] = make_uint4(__float_as_uint(oac[i]), __float_as_uint(olm[i]), (unsigned) orec[i], oek[i]); }
          };
          auto park_load = [&](const int kb, float* oac, float* olm, int* orec, unsigned* oek) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < kB; i++) {
              int c = (kb + i) * nthr + tq; c = (c < C) ? c : kFastK * nthr;                // K > kFastK: that slot exists
              const uint4 v = ovf[c - kFastK * nthr];
              oac[i] = __uint_as_float(v.x); olm[i] = __uint_as_float(v.y); orec[i] = (int) v.z; oek[i] = v.w;
            }
          };
          auto runP3 = [&](auto NOPEN) __attribute__((always_inline)) {
#pragma unroll
            for (int g8 = 0; g8 < kFastK; g8 += kB) if (g8 < K) expand8(NOPEN, g8, &qac[g8], &qlm[g8], &qrec[g8], &ek[g8]);
            for (int kb = kFastK; kb < K; kb += kB) {
              float oac[kB], olm[kB]; int orec[kB]; unsigned oek[kB];
              expand8(NOPEN, kb, oac, olm, orec, oek); park_store(kb, oac, olm, orec, oek);
            }
          };
          if (ka->D.noPen) runP3(std::true_type{}); else runP3(std::false_type{});
          TICK(2);
          locMin = wave_min_d_uni(locMin); locMag = wave_max_f_uni(locMag);
          if (lq == 0) { s_waveMin[wq] = locMin; s_waveMag[wq] = locMag; }
          TICK(3);
          __syncthreads();
          TICK(4);
          for (int i = tq; i < kFastC / 32; i += nthr) s_bm[i] = 0u;          // every slot has found its token: the bitmap is free for the layout of the next frame (P6)
          topScore = HUGE_VAL;
          float frameMag;
          { const double v = (lq < nw) ? s_waveMin[lq] : HUGE_VAL; const float g = (lq < nw) ? s_waveMag[lq] : 0.0f;
            topScore = wave_min_d_uni(v); frameMag = wave_max_f_uni(g); }
          // A token whose score is above (this frame's best emitting total + beam) fails the beam test of the next frame
          // (decoder.h:586-588) and is never looked at again: it is counted (activeHypos, maxActive) but neither written to the list
          // nor to the back-pointer arena.  The order of the tokens that stay is unchanged, so the next frame's arrival slots are too.
          // Not on the last frame (the end expansion takes every token) and not when the lists are dumped.
          RELOAD();
          const double threshNext = __dadd_rn(topScore, ka->D.beam);
          const bool prune = !dump && (fr + 1 < T);
          // A LATER arrival that far above the threshold cannot touch a token that stays: a token that stays has score <= threshNext, and its
          // unrounded total (what the recombination compares, decoder.h:519-528) is within 2^-23 (|ac| + |lm|) of its score, so it beats
          // both such an arrival and whatever incumbent that arrival would have replaced (whose score is above the arrival's total).  Which of
          // two tokens above the threshold a state ends with is not observable (neither is written).  Such arrivals skip the chain: no side
          // record, no replay -- about two thirds of the later arrivals at the usual beam.
          const double doomT = prune ? __dadd_rn(threshNext, (double) __fmul_rn(frameMag, 4.76837158203125e-07f)) : HUGE_VAL;   // gap = 2^-21 x bound
          // ---- P4/P5, once per pass over the state table: later arrivals hang themselves on their bucket (the key word becomes
          // the chain head), then every first arrival folds its chain in slot order (decoder.h:519-528)
          unsigned long long firstMask = 0ull;
          auto later8 = [&](const int g8, const int pass, const float* ac8, const float* lm8, const int* rec8, const unsigned* ek8) __attribute__((always_inline)) {
            double tt[kB]; uint32_t pb[kB];
#pragma unroll
            for (int i = 0; i < kB; i++) {                                      // what a later arrival hands over; loaded for all (coalesced, in flight together)
              int c = (g8 + i) * nthr + tq; c = (c < C) ? c : 0;
              tt[i] = ttlS[c]; pb[i] = ctk[ek8[i] & 0x1FFFu].bp;
            }
#pragma unroll
            for (int i = 0; i < kB; i++) {
              const int c = (g8 + i) * nthr + tq;
              if (c < C && (ek8[i] & (1u << 28)) && (int) ((ek8[i] >> 27) & 1u) == pass && !((firstMask >> (g8 + i)) & 1ull)) {
                const unsigned h = (ek8[i] >> 13) & 0x3FFFu;
                if (hfirst[h] == (unsigned) c) firstMask |= 1ull << (g8 + i);
                else if (!((double) __fadd_rn(ac8[i], lm8[i]) > doomT)) {
                  const int sx = atomicAdd(&s_sideN, 1);
                  const unsigned nx = atomicExch(&hkey[h], 0x80000000u | (unsigned) sx);
                  Side sd; sd.ttl = tt[i]; sd.ac = ac8[i]; sd.lm = lm8[i]; sd.rec = rec8[i]; sd.prevBp = pb[i]; sd.c = c; sd.next = nx;
                  if (sx < sideLds) sideL[sx] = sd; else side[sx] = sd;
                }
              }
            }
          };
          // (a first arrival that has been folded carries neither bucket nor pass bit any more: ek = token index, or bit31 | side index)
          // The replay of a state with two or more later arrivals: next replacement = smallest later slot that beats the incumbent,
          // until none does.  Returns the winning side record or -1.
          auto fold_multi = [&](const int wslot0, const unsigned head, const float ac, const float lm) __attribute__((always_inline)) -> int {
            int wslot = wslot0, wIdx = -1; double fw = (double) __fadd_rn(ac, lm);
            for (;;) {
              int best = 0x7FFFFFFF, bi = -1, steps = 0;
              for (unsigned p = head; (p & 0x80000000u) && steps <= C; steps++) {
                const int pi = (int) (p & 0x7FFFFFFFu);
                int pc; double pt; unsigned pn;
                side_link(sideL, side, sideLds, pi, pc, pt, pn);
                if (pc > wslot && pc < best && pt < fw) { best = pc; bi = pi; }
                p = pn;
              }
              if (bi < 0) break;
              wslot = best; wIdx = bi;
              fw = (double) side_score(sideL, side, sideLds, bi);
            }
            return wIdx;
          };
          // One placement's fold.  defer == nullptr: everything in place.  Otherwise a chain of two or more later arrivals is left for
          // the deferred round (its head goes to *defer): there the lanes of a wave replay their chains side by side instead of one
          // placement slot after the other with a single lane at work.
          auto fold1 = [&](const int k, const int pass, float& ac, float& lm, int& rec, unsigned& ekk, unsigned* defer) __attribute__((always_inline)) {
            const bool mine = ((firstMask >> k) & 1ull) && !(ekk & 0x80000000u) && (ekk & (1u << 28)) && (int) ((ekk >> 27) & 1u) == pass;
            if (mine) {
              const unsigned head = hkey[(ekk >> 13) & 0x3FFFu];
              ekk &= 0x1FFFu;
              if (head & 0x80000000u) {
                int wIdx = -1;
                const int hi = (int) (head & 0x7FFFFFFFu);
                unsigned hnext; double pt;
                { int cdummy; side_link(sideL, side, sideLds, hi, cdummy, pt, hnext); }
                if (!(hnext & 0x80000000u)) {                                  // one later arrival (the usual case): a single comparison
                  if (pt < (double) __fadd_rn(ac, lm)) wIdx = hi;
                } else if (defer) *defer = head;
                else wIdx = fold_multi(k * nthr + tq, head, ac, lm);
                if (wIdx >= 0) {
                  side_winner(sideL, side, sideLds, wIdx, ac, lm, rec);
                  ekk = 0x80000000u | (unsigned) wIdx;
                }
              }
            }
          };
          // second pass: the waiting half of the states enters the (wiped) table
          auto insert8 = [&](const int g8, const int* rec8, unsigned* ek8) __attribute__((always_inline)) {
            int xd[kB];
#pragma unroll
            for (int i = 0; i < kB; i++) xd[i] = ka->G.xrecD[rec8[i] & 0x3FFFFFFF].dst;
#pragma unroll
            for (int i = 0; i < kB; i++) {
              const int c = (g8 + i) * nthr + tq;
              if (c < C && (ek8[i] & (1u << 27)) && !(ek8[i] & 0x80000000u)) ek8[i] |= (table_insert((unsigned) xd[i], (unsigned) xd[i] * 2654435761u, c) << 13) | (1u << 28);
            }
          };
          for (int pass = 0; pass < nPass; pass++) {
            if (pass > 0) {
              __syncthreads();                                                 // every chain of the pass before has been folded
              { uint4* h4 = reinterpret_cast<uint4*>(hkey); const int q4 = hashN >> 2;
                for (int i = tq; i < 2 * q4; i += nthr) { unsigned wv = (i < q4) ? 0u : 0xFFFFFFFFu; asm volatile("" : "+v"(wv)); h4[i] = make_uint4(wv, wv, wv, wv); }   /* (opaque: a hoisted constant vector lived through the whole frame loop, partly in scratch) */ }
              __syncthreads();
#pragma unroll
              for (int g8 = 0; g8 < kFastK; g8 += kB) if (g8 < K) insert8(g8, &qrec[g8], &ek[g8]);
              for (int kb = kFastK; kb < K; kb += kB) {
                float oac[kB], olm[kB]; int orec[kB]; unsigned oek[kB]; park_load(kb, oac, olm, orec, oek);
                insert8(kb, orec, oek); park_store(kb, oac, olm, orec, oek);
              }
              __syncthreads();
            }
            {
              // the status of every register slot first (one table read each, all in flight), then the few later arrivals that matter hang
              // themselves on their buckets side by side: a lane works through its own, whatever their slot number, and only they fetch their
              // unrounded total and the parent's back pointer from memory (fetching them for every slot cost a memory round trip per batch)
              unsigned later = 0u;
#pragma unroll
              for (int k = 0; k < kFastK; k++) {
                const bool cand = k < K && k * nthr + tq < C && (ek[k] & (1u << 28)) && (int) ((ek[k] >> 27) & 1u) == pass && !((firstMask >> k) & 1ull);
                if (cand) {
                  const unsigned f1 = hfirst[(ek[k] >> 13) & 0x3FFFu];
                  if (f1 == (unsigned) (k * nthr + tq)) firstMask |= 1ull << k;
                  else if (!((double) __fadd_rn(qac[k], qlm[k]) > doomT)) later |= 1u << k;
                }
              }
              asm volatile("" : "+v"(later));
              __builtin_amdgcn_sched_barrier(0);
              while (__any(later != 0u)) {
                if (later) {
                  const int k = __ffs((int) later) - 1; later &= later - 1u;
                  unsigned ekk = ek[0];
#pragma unroll
                  for (int j = 1; j < kFastK; j++) ekk = (k == j) ? ek[j] : ekk;
                  const int c = k * nthr + tq;
                  const double tt = ttlS[c]; const uint32_t pb = ctk[ekk & 0x1FFFu].bp;
                  const unsigned h = (ekk >> 13) & 0x3FFFu;
                  const int sx = atomicAdd(&s_sideN, 1);
                  const unsigned nx = atomicExch(&hkey[h], 0x80000000u | (unsigned) sx);
                  float ac = qac[0], lm = qlm[0]; int rec = qrec[0];
#pragma unroll
                  for (int j = 1; j < kFastK; j++) { const bool is = (k == j); ac = is ? qac[j] : ac; lm = is ? qlm[j] : lm; rec = is ? qrec[j] : rec; }
                  Side* dstS = (sx < sideLds) ? nullptr : &side[sx];
                  if (!dstS) { Side& q = sideL[sx]; q.ttl = tt; q.ac = ac; q.lm = lm; q.rec = rec; q.prevBp = pb; q.c = c; q.next = nx; }
                  else { dstS->ttl = tt; dstS->ac = ac; dstS->lm = lm; dstS->rec = rec; dstS->prevBp = pb; dstS->c = c; dstS->next = nx; }
                }
              }
            }
            for (int kb = kFastK; kb < K; kb += kB) { float oac[kB], olm[kB]; int orec[kB]; unsigned oek[kB]; park_load(kb, oac, olm, orec, oek); later8(kb, pass, oac, olm, orec, oek); }
            if (pass == 0) TICK(20);
            __syncthreads();
            if (pass == 0) TICK(5);
            {
              unsigned pend = 0u, dh[kFastK];
#pragma unroll
              for (int k = 0; k < kFastK; k++) { dh[k] = 0u; if (k < K) { fold1(k, pass, qac[k], qlm[k], qrec[k], ek[k], &dh[k]); if (dh[k]) pend |= 1u << k; } }
              if (pass == 0) TICK(24);
              while (__any(pend != 0u)) {                                     // deferred round(s): one chain per lane at a time
                if (pend) {
                  const int k = __ffs((int) pend) - 1; pend &= pend - 1u;
                  unsigned head = dh[0]; float ac = qac[0], lm = qlm[0];
#pragma unroll
                  for (int j = 1; j < kFastK; j++) { const bool is = (k == j); head = is ? dh[j] : head; ac = is ? qac[j] : ac; lm = is ? qlm[j] : lm; }
                  const int wIdx = fold_multi(k * nthr + tq, head, ac, lm);
                  if (wIdx >= 0) {
                    float wac, wlm; int wrec;
                    side_winner(sideL, side, sideLds, wIdx, wac, wlm, wrec);
#pragma unroll
                    for (int j = 0; j < kFastK; j++) if (k == j) { qac[j] = wac; qlm[j] = wlm; qrec[j] = wrec; ek[j] = 0x80000000u | (unsigned) wIdx; }
                  }
                }
              }
            }
            if (pass == 0) TICK(25);
            for (int kb = kFastK; kb < K; kb += kB) {
              float oac[kB], olm[kB]; int orec[kB]; unsigned oek[kB]; park_load(kb, oac, olm, orec, oek);
#pragma unroll
              for (int i = 0; i < kB; i++) if (kb + i < K) fold1(kb + i, pass, oac[i], olm[i], orec[i], oek[i], nullptr);
              park_store(kb, oac, olm, orec, oek);
            }
          }
          TICK(12);
          if (s_err) { status = DSR_E_ALLOCATION; break; }                     // (uniform: written before the barriers above)
          unsigned long long keepMask = 0ull;
#pragma unroll
          for (int k = 0; k < kFastK; k++) if (k < K && ((firstMask >> k) & 1ull)) {
            const float sc1 = __fadd_rn(qac[k], qlm[k]);
            if (!prune || !((double) sc1 > threshNext)) keepMask |= 1ull << k;
          }
          for (int kb = kFastK; kb < K; kb += kB) {
            float oac[kB], olm[kB]; int orec[kB]; unsigned oek[kB]; park_load(kb, oac, olm, orec, oek);
#pragma unroll
            for (int i = 0; i < kB; i++) if (kb + i < K && ((firstMask >> (kb + i)) & 1ull)) {
              const float sc1 = __fadd_rn(oac[i], olm[i]);
              if (!prune || !((double) sc1 > threshNext)) keepMask |= 1ull << (kb + i);
            }
          }
          // count the tokens that stay per (k, wave) group of 64 slots, and all new tokens for the statistics
          int allFirst = 0;
          for (int k = 0; k < K; k++) {
            const unsigned long long bal = __ballot((keepMask >> k) & 1ull);
            if (lq == 0) s_cnt[k * nw + wq] = __popcll(bal);
            allFirst += __popcll(__ballot((firstMask >> k) & 1ull));
          }
          if (lq == 0) s_waveTotE[wq] = allFirst;
          TICK(13);
          __syncthreads();
          TICK(14);
          {                                                                    // every chain has been folded: the state table is dead.  Wiped here, under wave 0's prefix sum
            uint4* h4 = reinterpret_cast<uint4*>(hkey); const int q4 = hashN >> 2;
            for (int i = tq; i < 2 * q4; i += nthr) { unsigned wv = (i < q4) ? 0u : 0xFFFFFFFFu; asm volatile("" : "+v"(wv)); h4[i] = make_uint4(wv, wv, wv, wv); }
          }
          if (wq == 0) {                                                       // exclusive prefix over (k, wave) = slot order of the groups
            const int nG = K * nw;                                             // <= 6 * 64
            int a[6], tot = 0;
#pragma unroll
            for (int q = 0; q < 6; q++) { a[q] = (6 * lq + q < nG) ? s_cnt[6 * lq + q] : 0; tot += a[q]; }
            const int incl = wave_incl_scan(tot, lq); int run = incl - tot;
#pragma unroll
            for (int q = 0; q < 6; q++) { if (6 * lq + q < nG) s_cnt[6 * lq + q] = run; run += a[q]; }
            if (lq == 63) s_waveTot[0] = incl;
          }
          __syncthreads();
          TICK(6);
          numNew = uni(s_waveTot[0]);
          { const int a = (lq < nw) ? s_waveTotE[lq] : 0; numStat = __builtin_amdgcn_readlane(wave_incl_scan(a, lq), 63); }
          RELOAD();
          if (numNew > ka->D.maxTok || (segS > 0 ? arenaUsed : arenaOff) + numNew > ka->D.arenaCap) { status = DSR_E_ALLOCATION; break; }
          if (segS > 0 && arenaOff + numNew > chunkEnd) {                      // (uniform) the utterance's next run of back-pointer records from the pool
            const long need = numNew > ka->D.poolChunk ? (long) numNew : ka->D.poolChunk;
            __syncthreads();
            if (tid == 0) s_chunk = (long long) atomicAdd(ka->D.poolNext, (unsigned long long) need);
            __syncthreads();
            arenaOff = (long) s_chunk; chunkEnd = arenaOff + need;
            if (chunkEnd > ka->D.poolCap) { status = DSR_E_ALLOCATION; break; }
          }
          const XRecD* const xrecW = ka->G.xrecD; TokA* const outA = nxtA; TokB* const outB = nxtB; Bp* const outBp = arena + arenaOff;
          // ---- P6: the new list in reverse first-arrival order + back pointers; the state table is wiped for the next frame
          auto write4 = [&](auto NB, const int g4, const float* ac4, const float* lm4, const int* rec4, const unsigned* ek4) __attribute__((always_inline)) {
            constexpr int nb = decltype(NB)::value;
            int4 dx[nb]; uint32_t pv[nb];
#pragma unroll
            for (int i = 0; i < nb; i++) {                                      // unconditional loads (every index is in bounds), in flight together;
              const bool kp = (keepMask >> (g4 + i)) & 1ull;                   // placements that are not written all read record 0 / token 0 (one line)
              dx[i] = *reinterpret_cast<const int4*>(&xrecW[kp ? (rec4[i] & 0x3FFFFFFF) : 0].dst);       // same state for every arrival
              const bool sw = kp && (ek4[i] & 0x80000000u) != 0u; const unsigned si = ek4[i] & 0x7FFFFFFFu;
              const uint32_t* pb = (sw && si >= (unsigned) sideLds) ? &side[si].prevBp : &ctk[(sw || !kp) ? 0u : (ek4[i] & 0x1FFFu)].bp;
              pv[i] = *pb;
              if (sw && si < (unsigned) sideLds) pv[i] = sideL[si].prevBp;
            }
#pragma unroll
            for (int i = 0; i < nb; i++) {
              const int k = g4 + i;
              if (k < K) {
                const bool isFirst = (keepMask >> k) & 1ull;
                const unsigned long long bal = __ballot(isFirst);
                if (isFirst) {
                  const int pos = numNew - 1 - (s_cnt[k * nw + wq] + __popcll(bal & ((1ull << lq) - 1ull)));
                  TokA na; na.ac = ac4[i]; na.lm = lm4[i]; na.bp = (uint32_t) (arenaOff + pos); na.xs = (uint32_t) dx[i].z | ((rec4[i] & 0x40000000) ? 0x80000000u : 0u);
                  TokB nb; nb.node = dx[i].x; nb.cnt = dx[i].w;
                  Bp bp; bp.prev = pv[i]; bp.rec = (uint32_t) (rec4[i] & 0x3FFFFFFF);
                  outA[pos] = na; outB[pos] = nb; outBp[pos] = bp;
                  if (pos < cntCap) cntL[pos] = (unsigned short) dx[i].w;
                }
              }
            }
          };
          // Three placements in four are not written (pruned, or later arrivals): the ones that are go through the (wiped) state table as a dense
          // list -- {ac, lm, record, where the parent's back pointer is} at its rank -- and the loads and stores of the write run over that list,
          // every lane at work, instead of over all slots with a quarter of the lanes.  A thread restores the table words it has read.
          const bool dense = numNew <= (hashN >> 1);
          if (dense) {
            uint4* const stage = reinterpret_cast<uint4*>(hkey);
            auto stage1 = [&](const int k, const float ac, const float lm, const int rec, const unsigned ekk) __attribute__((always_inline)) {
              const bool isFirst = (keepMask >> k) & 1ull;
              const unsigned long long bal = __ballot(isFirst);
              if (isFirst) stage[s_cnt[k * nw + wq] + __popcll(bal & ((1ull << lq) - 1ull))] = make_uint4(__float_as_uint(ac), __float_as_uint(lm), (unsigned) rec, ekk);
            };
#pragma unroll
            for (int k = 0; k < kFastK; k++) if (k < K) stage1(k, qac[k], qlm[k], qrec[k], ek[k]);
            for (int kb = kFastK; kb < K; kb += kB) {
              float oac[kB], olm[kB]; int orec[kB]; unsigned oek[kB]; park_load(kb, oac, olm, orec, oek);
#pragma unroll
              for (int i = 0; i < kB; i++) if (kb + i < K) stage1(kb + i, oac[i], olm[i], orec[i], oek[i]);
            }
            __syncthreads();
            TICK(21);
            const int q4 = hashN >> 2;
            const bool layNext = prune && numNew <= 2 * nthr;                  // (one round of the loop below)
            preC = -1;
            for (int f0 = 0; f0 < numNew; f0 += 2 * nthr) {
              uint4 en[2]; int4 dx[2]; uint32_t pv[2];
#pragma unroll
              for (int i = 0; i < 2; i++) {
                const int f = f0 + i * nthr + tq; const bool on = f < numNew;
                en[i] = stage[on ? f : 0];
                if (on) { unsigned wv = (f < q4) ? 0u : 0xFFFFFFFFu; asm volatile("" : "+v"(wv)); stage[f] = make_uint4(wv, wv, wv, wv); }
              }
#pragma unroll
              for (int i = 0; i < 2; i++) {
                const bool on = f0 + i * nthr + tq < numNew;
                dx[i] = *at32<int4>(xrecW, (on ? (en[i].z & 0x3FFFFFFFu) : 0u) * 32u + 16u);
                const bool sw = on && (en[i].w & 0x80000000u) != 0u; const unsigned si = en[i].w & 0x7FFFFFFFu;
                const uint32_t* pb = (sw && si >= (unsigned) sideLds) ? &side[si].prevBp : &ctk[(sw || !on) ? 0u : (en[i].w & 0x1FFFu)].bp;
                pv[i] = *pb;
                if (sw && si < (unsigned) sideLds) pv[i] = sideL[si].prevBp;
              }
              // The list this loop writes is the list the next frame expands, every token of it (pruning on: all pass the beam).  A token's first slot
              // there is the sum of the expansion counts of the tokens before it in the list = after it in rank: one block-wide sum over the counts the
              // loop has in hand anyway gives every token its slot offset, and the bitmap of run starts and the per-group table -- what P1 and P2 of the
              // next frame would rebuild from the list -- are written here.  That frame then starts at P3 (one barrier for its score row).
              if (layNext) {
                const int cA0 = (tq < numNew) ? dx[0].w : 0, cB0 = (nthr + tq < numNew) ? dx[1].w : 0;
                const int inA = wave_incl_scan(cA0, lq), inB = wave_incl_scan(cB0, lq);
                const bool bad = (tq < numNew && cA0 == 0) || (nthr + tq < numNew && cB0 == 0);     // a token without arcs: the list is not its own compact list
                const int badW = __any(bad) ? 1 : 0;
                if (lq == 63) { s_waveTot[wq] = inA; s_waveTotE[wq] = inB | (badW << 30); }
                __syncthreads();
                const int ta = (lq < nw) ? s_waveTot[lq] : 0, tbv = (lq < nw) ? s_waveTotE[lq] : 0, tb = tbv & 0x3FFFFFFF;
                const int sa = wave_incl_scan(ta, lq), sb = wave_incl_scan(tb, lq);
                const int totA = __builtin_amdgcn_readlane(sa, 63), totB = __builtin_amdgcn_readlane(sb, 63), wu = uni(wq);
                const int total = totA + totB;
                const int offA = total - (__builtin_amdgcn_readlane(sa - ta, wu) + inA), offB = total - (totA + __builtin_amdgcn_readlane(sb - tb, wu) + inB);
                const bool anyBad = __any((tbv >> 30) & 1);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                  const int f = i * nthr + tq; const int off = i ? offB : offA, cn = i ? cB0 : cA0;
                  if (f < numNew && cn > 0) {
                    const int e = numNew - 1 - f;
                    atomicOr(&s_bm[off >> 5], 1u << (off & 31));
                    for (int g = (off >> 6) + 1; g <= ((off + cn) >> 6); g++) s_gbase[g] = (unsigned) (e + 1) | ((unsigned) (64 * g - off) << 13);
                  }
                }
                if (tq == 0) { s_sideN = 0; s_gbase[0] = 0; s_err = 0; }
                preC = anyBad ? -1 : total;
              }
#pragma unroll
              for (int i = 0; i < 2; i++) {
                const int f = f0 + i * nthr + tq;
                if (f < numNew) {
                  const int pos = numNew - 1 - f;
                  TokA na; na.ac = __uint_as_float(en[i].x); na.lm = __uint_as_float(en[i].y); na.bp = (uint32_t) (arenaOff + pos); na.xs = (uint32_t) dx[i].z | ((en[i].z & 0x40000000u) ? 0x80000000u : 0u);
                  TokB nb; nb.node = dx[i].x; nb.cnt = dx[i].w;
                  Bp bp; bp.prev = pv[i]; bp.rec = (uint32_t) (en[i].z & 0x3FFFFFFFu);
                  outA[pos] = na; outB[pos] = nb; outBp[pos] = bp;
                  if (pos < cntCap) cntL[pos] = (unsigned short) dx[i].w;
                }
              }
            }
            TICK(22);
          } else
          {
#pragma unroll
          for (int g4 = 0; g4 < kFastK; g4 += kW) if (g4 < K) write4(std::integral_constant<int, kW>{}, g4, &qac[g4], &qlm[g4], &qrec[g4], &ek[g4]);
          for (int kb = kFastK; kb < K; kb += kB) {
            float oac[kB], olm[kB]; int orec[kB]; unsigned oek[kB]; park_load(kb, oac, olm, orec, oek);
            write4(std::integral_constant<int, kB>{}, kb, oac, olm, orec, oek);
          }
          TICK(21);
          TICK(22);
          preC = -1;
          }
          cntOK = prune && numNew <= cntCap;
        }
      }
      if (!fast) {
      cntOK = false; preC = -1;
      // ======================= memory path =======================
      RELOAD();
      int* tokOff = ka->D.tokOff + (size_t) slot * (ka->D.maxTok + 1);
      int* tokCnt = ka->D.tokCnt + (size_t) slot * (ka->D.maxTok + 1);
      int* chead = ka->D.chead + (size_t) slot * ka->D.maxCand;
      int* owner = ka->D.owner + (size_t) slot * ka->D.maxCand;
      int* rank = ka->D.rank + (size_t) slot * ka->D.maxCand;
      unsigned* first = ka->D.first + (size_t) slot * ka->G.nNodes;
      unsigned tag = s_tag;
      __syncthreads();                                                         // (every thread has read the tag before thread 0 replaces it)
      if (tag <= 1u) { for (int i = tid; i < ka->G.nNodes; i += nthr) first[i] = 0xFFFFFFFFu; tag = 255u; __syncthreads(); } else tag--;
      if (tid == 0) s_tag = tag;
      const unsigned tagw = tag << 24;
      const bool sorted = EXTRA && ka->D.topN > 0 && mode == 0 && fr > 0;
      if (sorted) {
        // SortedIterator (decoder.h:298-320): the list sorted by the tokens' float scores (ties: list order -- std::sort leaves them unspecified),
        // of which _processFrame expands the first topN (:571-581).  Rank by counting: the lists of this mode are short (topN tokens' expansions).
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(cA);
        for (int i = tid; i < n; i += nthr) { const TokA t = curA[i]; keys[i] = ((unsigned long long) f2ord(__fadd_rn(t.ac, t.lm)) << 32) | (unsigned) i; }
        __syncthreads();
        for (int i = tid; i < n; i += nthr) {
          const unsigned long long ki = keys[i]; int rk = 0;
          for (int j = 0; j < n; j++) rk += (keys[j] < ki) ? 1 : 0;
          if (rk < ka->D.topN) { sprA[rk] = curA[i]; sprB[rk] = curB[i]; }
        }
        __syncthreads();
        { const int t = bufCur; bufCur = bufSpr; bufSpr = t; }
        n = n < ka->D.topN ? n : ka->D.topN;
      }
      const double threshM = (EXTRA && ka->D.topN > 0) ? HUGE_VAL : thresh;       // topN mode: no beam
      // ---------------- phase A: per-token placement counts, wave-local exclusive scan
      const int chunkT = ((n + nw * 64 - 1) / (nw * 64)) * 64;
      {
        int running = 0;
        const int b0 = wave * chunkT, b1 = (b0 + chunkT < n) ? b0 + chunkT : n;
        for (int base = b0; base < b1; base += 64) {
          const int i = base + lane; int cnt = 0;
          if (i < b1) {
            const TokA ta = curA[i]; const TokB tb = curB[i];
            if (mode == 0) {
              const float s = __fadd_rn(ta.ac, ta.lm);
              if (!((double) s > threshM)) cnt = tb.cnt;                       // beam (decoder.h:586-588)
            } else cnt = (ka->G.nodeFinal[tb.node] ? 1 : 0) + (ka->G.eoff[tb.node + 1] - ka->G.eoff[tb.node]);
          }
          const int incl = wave_incl_scan(cnt, lane);
          if (i < b1) { tokOff[i] = running + incl - cnt; tokCnt[i] = cnt; }
          running += __shfl(incl, 63, 64);
        }
        if (lane == 0) s_waveTot[wave] = running;
      }
      __syncthreads();
      int C = 0;
      for (int w = 0; w < nw; w++) C += s_waveTot[w];
      if (C > ka->D.maxCand) { status = DSR_E_ALLOCATION; break; }
      if (EXTRA && ka->D.latOn && s_latOff + C > ka->D.latCap) { status = DSR_E_ALLOCATION; break; }
      Cfr = C;
      const bool useHash = hashN > 0 && C <= (hashN >> 1) + (hashN >> 2);        // load factor <= 0.75 even if every placement is a new state
      // ---------------- phase A2: absolute offsets + owner fill
      for (int i = tid; i < n; i += nthr) {
        const int cnt = tokCnt[i];
        if (cnt == 0) continue;
        const int w = i / chunkT; int base = 0;
        for (int q = 0; q < w; q++) base += s_waveTot[q];
        const int off = base + tokOff[i];
        tokOff[i] = off;
        for (int j = 0; j < cnt; j++) owner[off + j] = i;
      }
      __syncthreads();
      // ---------------- phase B: one thread per placement
      double locMin = HUGE_VAL;
      for (int c = tid; c < C; c += nthr) {
        const int i = owner[c]; const TokA ta = curA[i]; const int nd = curB[i].node; const bool tokSil = (ta.xs >> 31) != 0u;
        Tok t; t.node = nd; t.ac = ta.ac; t.lm = ta.lm; t.bp = ta.bp;
        const int j = c - tokOff[i];
        double ac = (double) t.ac, lm; int dst, recId; bool silArc = false;
        if (mode == 0) {
          recId = (int) (ta.xs & 0x7FFFFFFFu) + j; const XRec x = ka->G.xrec[recId];
          const int plen = (int) (x.meta & 0xFFFFu);
          double lmNode = (double) t.lm; uint32_t prevIn = tokSil ? ka->D.silenceX : (ka->D.silenceX + 1u);   // only equality with silenceX matters
          bool prevNull = (t.bp == kNone) && (fr == 0);
          if (plen) {
            const int* pp = ka->G.path + ka->G.xpathOff[recId];
            for (int h = 0; h < plen; h++) {                                   // intermediate epsilon tokens (decoder.h:979-983)
              const int a = pp[h];
              double l = __dadd_rn(lmNode, __dmul_rn(ka->D.lmScale, (double) ka->G.arcCost[a]));
              if (ka->G.arcOut[a] != 0) l = __dadd_rn(l, __dmul_rn(ka->D.lmScale, ka->D.lmPenalty));
              if (0u == ka->D.silenceX && (prevNull || prevIn != ka->D.silenceX)) l = __dadd_rn(l, __dmul_rn(ka->D.lmScale, ka->D.silPenalty));
              lmNode = (double) (float) l; prevIn = 0u; prevNull = false;
            }
          }
          lm = __dadd_rn(lmNode, __dmul_rn(ka->D.lmScale, (double) x.cost));
          if (x.meta & 0x10000u) lm = __dadd_rn(lm, __dmul_rn(ka->D.lmScale, ka->D.lmPenalty));
          if ((uint32_t) (x.dist + 1) == ka->D.silenceX && (prevNull || prevIn != ka->D.silenceX)) lm = __dadd_rn(lm, __dmul_rn(ka->D.lmScale, ka->D.silPenalty));
          ac = __dadd_rn(ac, (double) (useLdsRow ? srow[x.dist] : rowG[x.dist]));
          dst = x.dst; silArc = ((uint32_t) (x.dist + 1) == ka->D.silenceX);
        } else {
          const int hasSelf = ka->G.nodeFinal[nd] ? 1 : 0;
          if (hasSelf && j == 0) {                                             // _expandToEnd self placement (decoder.h:506-509)
            const float lmf = (float) __dadd_rn((double) t.lm, __dmul_rn(ka->D.lmScale, (double) ka->G.nodeCost[nd]));
            lm = (double) lmf; dst = nd; recId = (int) 0x7FFFFFFE;
          } else {
            recId = ka->G.eoff[nd] + (j - hasSelf); const ERec e = ka->G.erec[recId];
            const int* pp = ka->G.path + e.pathOff; double lmNode = (double) t.lm; double l = lmNode;
            uint32_t prevIn = tokSil ? ka->D.silenceX : (ka->D.silenceX + 1u);
            for (int h = 0; h < e.pathLen; h++) {                              // _expandNodeToEnd (decoder.h:992-1015)
              const int a = pp[h];
              l = __dadd_rn(lmNode, __dmul_rn(ka->D.lmScale, (double) ka->G.arcCost[a]));
              if (ka->G.arcOut[a] != 0) l = __dadd_rn(l, __dmul_rn(ka->D.lmScale, ka->D.lmPenalty));
              if (0u == ka->D.silenceX && prevIn != ka->D.silenceX) l = __dadd_rn(l, __dmul_rn(ka->D.lmScale, ka->D.silPenalty));
              lmNode = (double) (float) l; prevIn = 0u;
            }
            lm = __dadd_rn(l, __dmul_rn(ka->D.lmScale, (double) ka->G.nodeCost[e.lastSrc]));
            dst = e.dst; recId = (int) ((uint32_t) recId | kEndBit);
          }
        }
        const double ttl = __dadd_rn(ac, lm);
        CandA a2; a2.ttl = ttl; a2.ac = (float) ac; a2.lm = (float) lm; cA[c] = a2;
        if (useHash) {
          unsigned h = ((unsigned) dst * 2654435761u) >> 7 & (unsigned) (hashN - 1);
          for (;;) {
            const unsigned k = atomicCAS(&hkey[h], 0u, (unsigned) dst + 1u);
            if (k == 0u || k == (unsigned) dst + 1u) break;
            h = (h + 1u) & (unsigned) (hashN - 1);
          }
          atomicMin(&hfirst[h], (unsigned) c);
          rank[c] = (int) h;
        } else atomicMin(&first[dst], tagw | (unsigned) c);
        chead[c] = -1;
        if (silArc) recId |= 0x40000000;
        CandB b2; b2.dst = dst; b2.next = -1; b2.rec = recId; b2.prevBp = t.bp; cB[c] = b2;
        if (mode == 0 && ttl < locMin) locMin = ttl;                           // _topScore (emitting placements only)
      }
      locMin = wave_min_d(locMin);
      if (lane == 0) s_waveMin[wave] = locMin;
      __syncthreads();
      topScore = HUGE_VAL;
      for (int w = 0; w < nw; w++) { const double v = s_waveMin[w]; if (v < topScore) topScore = v; }
      // ---------------- phase C0: later arrivals hang themselves on their state's first-arrival placement
      for (int c = tid; c < C; c += nthr) {
        int f;
        if (useHash) f = (int) hfirst[rank[c]];
        else { const int dst = cB[c].dst; f = (int) (ld_u32(&first[dst]) & 0x00FFFFFFu); }
        if (f != c) { const int nx = atomicExch(&chead[f], c); cB[c].next = nx; chead[c] = -2; }      // -2: not a first arrival
      }
      __syncthreads();
      if (useHash) for (int i = tid; i < 2 * hashN; i += nthr) hkey[i] = (i < hashN) ? 0u : 0xFFFFFFFFu;   // ready for the next frame
      // ---------------- phase C1: fold per destination state (by its first-arrival thread), count new tokens
      const int chunkC = ((C + nw * 64 - 1) / (nw * 64)) * 64;
      // (as on the register path: tokens above this frame's best emitting total + beam are counted but not written)
      const double threshNextM = __dadd_rn(topScore, ka->D.beam);
      const bool pruneM = !dump && mode == 0 && (fr + 1 < T) && !(EXTRA && ka->D.topN > 0);
      {
        int running = 0, runAll = 0;
        const int b0 = wave * chunkC, b1 = (b0 + chunkC < C) ? b0 + chunkC : C;
        for (int base = b0; base < b1; base += 64) {
          const int c = base + lane; bool isFirst = false, isAny = false;
          if (c < b1) {
            const int h0 = ld_i32(&chead[c]);
            isFirst = (h0 != -2);
            if (isFirst) {
              int w = c; CandA aw = cA[w];
              double fw = (double) __fadd_rn(aw.ac, aw.lm);                   // incumbent's float score()
              for (;;) {                                                       // next replacement = smallest later slot that beats it
                int best = 0x7FFFFFFF;
                int steps = 0;                                                 // the list has at most C nodes: never spin on a corrupt link
                for (int p = h0; p >= 0 && steps <= C; p = cB[p].next, steps++) if (p > w && p < best && cA[p].ttl < fw) best = p;
                if (best == 0x7FFFFFFF) break;
                w = best; aw = cA[w]; fw = (double) __fadd_rn(aw.ac, aw.lm);
              }
              isAny = true;
              if (pruneM && ((double) __fadd_rn(aw.ac, aw.lm) > threshNextM)) { isFirst = false; w = -3; }    // -3: a new token that is not written
              chead[c] = w;                                                    // winner of this state
            }
          }
          const unsigned long long bal = __ballot(isFirst);
          if (isFirst) rank[c] = running + __popcll(bal & ((1ull << lane) - 1ull));
          running += __popcll(bal); runAll += __popcll(__ballot(isAny));
        }
        if (lane == 0) { s_waveTot[wave] = running; s_waveTotE[wave] = runAll; }
      }
      __syncthreads();
      numStat = 0;
      for (int w = 0; w < nw; w++) { numNew += s_waveTot[w]; numStat += s_waveTotE[w]; }
      if (numNew > ka->D.maxTok || (segS > 0 ? arenaUsed : arenaOff) + numNew > ka->D.arenaCap) { status = DSR_E_ALLOCATION; break; }
      if (segS > 0 && arenaOff + numNew > chunkEnd) {
        const long need = numNew > ka->D.poolChunk ? (long) numNew : ka->D.poolChunk;
        __syncthreads();
        if (tid == 0) s_chunk = (long long) atomicAdd(ka->D.poolNext, (unsigned long long) need);
        __syncthreads();
        arenaOff = (long) s_chunk; chunkEnd = arenaOff + need;
        if (chunkEnd > ka->D.poolCap) { status = DSR_E_ALLOCATION; break; }
      }
      // ---------------- phase C2: write the new token list (reverse first-arrival order) + back pointers
      {
        int wbase = 0; for (int q = 0; q < wave; q++) wbase += s_waveTot[q];
        const int b0 = wave * chunkC, b1 = (b0 + chunkC < C) ? b0 + chunkC : C;
        for (int base = b0; base < b1; base += 64) {
          const int c = base + lane;
          if (c < b1) {
            const int w = chead[c];
            if (w >= 0) {
              const CandB bc = cB[c];
              const int pos = numNew - 1 - (wbase + rank[c]);
              const CandA aw = cA[w]; const CandB bw = (w == c) ? bc : cB[w]; const int recW = (bw.rec & 0x3FFFFFFF) | (bw.rec & (int) 0x80000000);
              const int xo = ka->G.xoff[bc.dst];
              TokA na; na.ac = aw.ac; na.lm = aw.lm; na.bp = (uint32_t) (arenaOff + pos); na.xs = (uint32_t) xo;
              TokB nb; nb.node = bc.dst; nb.cnt = ka->G.xoff[bc.dst + 1] - xo;
              Bp bp;
              if (mode == 0) {
                if (bw.rec & 0x40000000) na.xs |= 0x80000000u;
                bp.prev = bw.prevBp; bp.rec = (uint32_t) recW;
              } else {
                if (bw.rec == (int) 0x7FFFFFFE) { if (segS > 0) { const unsigned long long v = ld_u64_dev(&arena[bw.prevBp]); bp.prev = (uint32_t) v; bp.rec = (uint32_t) (v >> 32); } else { const Bp o = arena[bw.prevBp]; bp = o; } }     // replaces the token in its chain
                else { bp.prev = bw.prevBp; bp.rec = (uint32_t) bw.rec; }
              }
              nxtA[pos] = na; nxtB[pos] = nb; arena[arenaOff + pos] = bp;
              if (EXTRA && ka->D.latOn) ka->D.arenaLat[(size_t) u * ka->D.arenaCap + arenaOff + pos] = (int) (s_latOff + w);
            }
          }
        }
      }
      if (EXTRA && ka->D.latOn) {                                                 // this frame's placements, arrival order
        uint4* lat = ka->D.lat + (size_t) u * ka->D.latCap; double* ltt = ka->D.latTtl + (size_t) u * ka->D.latCap;
        const long latOff = (long) s_latOff;
        for (int c = tid; c < C; c += nthr) {
          const CandA a = cA[c]; const CandB b = cB[c];
          lat[latOff + c] = make_uint4(__float_as_uint(a.ac), __float_as_uint(a.lm), (unsigned) b.rec, b.prevBp); ltt[latOff + c] = a.ttl;
        }
        __syncthreads();                                                       // (every thread has read the offset before thread 0 advances it)
        if (tid == 0) { s_latOff = latOff + C; ka->D.latFrameOff[(size_t) u * (ka->Tmax + 3) + fr + 1] = latOff + C; }
      }
      }   // memory path
      __syncthreads();
      TICK(fast ? 7 : 8);
      RELOAD();
      if (mode == 0) {
        if (dump) {
          long* cnt = ka->D.dumpCount; const long o = cnt[0];
          if (o + numNew <= ka->D.dumpCap) {
            for (int i = tid; i < numNew; i += nthr) {
              const TokA t = ld_tok(&nxtA[i]); ka->D.dumpNode[o + i] = nxtB[i].node; ka->D.dumpAc[o + i] = t.ac; ka->D.dumpLm[o + i] = t.lm;
              ka->D.dumpArc[o + i] = ka->G.xarc[arena[t.bp].rec];
            }
          }
          __syncthreads();
          if (tid == 0) { ka->D.dumpFrameOff[fr] = o; ka->D.dumpFrameOff[fr + 1] = o + numNew; cnt[0] = o + numNew; cnt[1] = fr + 1; }
        }
        if (numStat < 0) numStat = numNew;                                     // memory path: every new token is written
        if (numStat == 0 || numNew == 0) { status = DSR_E_CONSISTENCY; break; } // no token can be expanded in the next frame: the reference never terminates from here
        { const int t = bufCur; bufCur = bufNxt; bufNxt = t; }
        n = numNew; arenaOff += numNew; arenaUsed += numNew;
        if (tid == nthr - 1) { s_stat[0] += numStat; s_stat[1] += Cfr; if (fast) s_stat[2] += 1; if (numStat > s_maxActive) s_maxActive = numStat; }   // (off wave 0's path: it carries the prefix sums)
        thresh = __dadd_rn(topScore, ka->D.beam);
      } else {
        // ---------------- best token (decoder.h:639-685): list order, strict '<' on the float score
        const TokA* lst = numNew > 0 ? nxtA : curA; const int cntL = numNew > 0 ? numNew : n;
        unsigned long long key = ~0ull;
        for (int i = tid; i < cntL; i += nthr) {
          const TokA t = lst[i]; const float s = __fadd_rn(t.ac, t.lm);
          if (s == s) { const unsigned long long k = ((unsigned long long) f2ord(s) << 32) | (unsigned) i; if (k < key) key = k; }
        }
        key = wave_min_u64(key);
        if (lane == 0) s_waveKey[wave] = key;
        if (EXTRA && ka->D.latOn) {                                               // _next after _expandToEnd (or _current when no token is final), list order
          const TokB* lstB = numNew > 0 ? nxtB : curB; int4* lf = ka->D.latFinal + (size_t) u * ka->D.maxTok;
          for (int i = tid; i < cntL; i += nthr) { const TokA t = ld_tok(&lst[i]); lf[i] = make_int4(lstB[i].node, (int) t.bp, __float_as_int(t.ac), __float_as_int(t.lm)); }
          if (tid == 0) { int* li = ka->D.latInfo + 4 * (size_t) u; li[0] = cntL; li[1] = numNew > 0 ? 1 : 0; li[2] = (int) (arenaOff + (numNew > 0 ? numNew : 0)); li[3] = T; }
        }
        __syncthreads();
        // traceback (bestHypo, decoder.h:748-773).  One thread follows the back pointers (one dependent 8-byte load per frame) and
        // leaves the hop records in scratch memory; everything else -- arcs per hop, their places in the list, the word sequence --
        // is done by the whole workgroup with two prefix sums.
        int* hopRec = reinterpret_cast<int*>(cB); const int hopCap = 2 * ka->D.maxCand; int* hopOff = hopRec + hopCap;
        // (the result record is written by thread 0 alone and lives in memory between its two steps: as a register struct of every thread it was
        // a block of zeros carried -- and spilled -- through the whole kernel)
        if (tid == 0) {
          dsr_decode_result* const res = ka->res;
          clear_result(&res[u]); dsr_decode_result& r = res[u];
          unsigned long long k = ~0ull; for (int w = 0; w < nw; w++) if (s_waveKey[w] < k) k = s_waveKey[w];
          r.frames = T - 1; r.reachedFinal = numNew > 0 ? 1 : 0; r.finalStatesN = numNew; /* every token of _next after _expandToEnd sits in a final state */ r.activeHypos = (long) s_stat[0]; r.placements = (long) s_stat[1] + Cfr; r.registerFrames = (long) s_stat[2]; r.maxActiveSeen = s_maxActive; r.status = DSR_OK;
          int nH = 0;
          if (k != ~0ull) {
            const TokA bt = ld_tok(&lst[(unsigned) (k & 0xFFFFFFFFu)]);
            r.ac = bt.ac; r.lm = bt.lm; r.score = __dadd_rn((double) bt.ac, (double) bt.lm);
            for (uint32_t bq = bt.bp; bq != kNone && nH < hopCap; ) {          // (time slicing: the records were written on other CUs -- read past this XCD's L2)
              uint2 e; if (segS > 0) { const unsigned long long v = ld_u64_dev(&arena[bq]); e.x = (unsigned) v; e.y = (unsigned) (v >> 32); } else e = *reinterpret_cast<const uint2*>(&arena[bq]);
              hopRec[nH++] = (int) e.y; bq = e.x;
            }
          } else r.status = DSR_E_CONSISTENCY;
          s_tb[0] = nH; s_tb[1] = 0;
        }
        __syncthreads();
        const int nH = s_tb[0];
        auto block_excl = [&](const int v, int& total) -> int {
          const int incl = wave_incl_scan(v, lane);
          if (lane == 63) s_waveTot[wave] = incl;
          __syncthreads();
          int base = 0, tot = 0;
          for (int w = 0; w < nw; w++) { const int q = s_waveTot[w]; if (w < wave) base += q; tot += q; }
          __syncthreads();
          total = tot; return base + incl - v;
        };
        int nA = 0;
        for (int b0 = 0; b0 < nH; b0 += nthr) {                                  // arcs per hop; hopOff = arcs of the hops walked before (= later in time)
          const int i = b0 + tid; int len = 0;
          if (i < nH) { const uint32_t rc = (uint32_t) hopRec[i]; len = (rc & kEndBit) ? ka->G.erec[rc & ~kEndBit].pathLen : (int) (ka->G.xrec[rc].meta & 0xFFFFu) + 1; }
          int total; const int ex = block_excl(len, total);
          if (i < nH) hopOff[i] = nA + ex;
          nA += total;
        }
        int* const arcsOut = ka->arcsOut; unsigned* const wordsOut = ka->wordsOut; const int maxPath = ka->maxPath; dsr_decode_result* const res = ka->res;
        int* ao = arcsOut ? arcsOut + (size_t) u * maxPath : nullptr;
        int nWloc = 0;
        for (int i = tid; i < nH; i += nthr) {                                   // every hop writes its arcs, first..last
          const uint32_t rc = (uint32_t) hopRec[i]; int pos = nA - hopOff[i];
          if (rc & kEndBit) {
            const ERec e = ka->G.erec[rc & ~kEndBit];
            for (int h = e.pathLen - 1; h >= 0; h--) { const int a = ka->G.path[e.pathOff + h]; pos--; if (ao && pos < maxPath) ao[pos] = a; if (ka->G.arcOut[a] != 0) nWloc++; }
          } else {
            const int a = ka->G.xarc[rc]; pos--; if (ao && pos < maxPath) ao[pos] = a; if (ka->G.arcOut[a] != 0) nWloc++;
            const int pl = (int) (ka->G.xrec[rc].meta & 0xFFFFu); const int po = ka->G.xpathOff[rc];
            for (int h = pl - 1; h >= 0; h--) { const int a2 = ka->G.path[po + h]; pos--; if (ao && pos < maxPath) ao[pos] = a2; if (ka->G.arcOut[a2] != 0) nWloc++; }
          }
        }
        if (nWloc) atomicAdd(&s_tb[1], nWloc);
        __syncthreads();
        if (wordsOut && ao) {                                                    // the words of the arc list, in order
          unsigned* wo = wordsOut + (size_t) u * maxPath; int q = 0;
          const int lim = nA < maxPath ? nA : maxPath;
          for (int b0 = 0; b0 < lim; b0 += nthr) {
            const int i = b0 + tid; unsigned o = 0u;
            if (i < lim) o = ka->G.arcOut[ao[i]];
            int total; const int ex = block_excl(o != 0u ? 1 : 0, total);
            if (o != 0u && q + ex < maxPath) wo[q + ex] = o;
            q += total;
          }
        }
        if (tid == 0) {
          if (res[u].status == DSR_OK) {
            res[u].nArcs = nA; res[u].nWords = s_tb[1];
            if (nH >= hopCap) res[u].status = DSR_E_ALLOCATION; else if (nA > maxPath && arcsOut) res[u].status = DSR_E_DIMENSION;
          }
        }
        __syncthreads();
        TICK(9);
      }
    }   // frames

    RELOAD();
    if (PROF && tid == 0) s_prof[15] = (long long) wall_clock64();
    if (segS > 0 && status == DSR_OK && fr <= T) {
      // the segment is done and the utterance is not: it is put down -- list and scalars to memory, visible to the whole device, THEN the segment count
      TokA* const svA = ka->D.saveA + (size_t) u * ka->D.maxTok; TokB* const svB = ka->D.saveB + (size_t) u * ka->D.maxTok;
      for (int i = tid; i < n; i += nthr) { svA[i] = curA[i]; svB[i] = curB[i]; }
      if (tid == 0) {
        SegState* const sst = ka->D.segState + u;
        st_i32(&sst->n, n); st_i32(&sst->status, DSR_OK); st_i32(&sst->maxActive, s_maxActive); st_u64_dev(&sst->arenaOff, (unsigned long long) arenaOff); st_u64_dev(&sst->chunkEnd, (unsigned long long) chunkEnd); st_u64_dev(&sst->arenaUsed, (unsigned long long) arenaUsed);
        st_u64_dev(&sst->thresh, (unsigned long long) __double_as_longlong(thresh));
        st_u64_dev(&sst->stat[0], (unsigned long long) s_stat[0]); st_u64_dev(&sst->stat[1], (unsigned long long) s_stat[1]); st_u64_dev(&sst->stat[2], (unsigned long long) s_stat[2]);
      }
      if (nq == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                // every thread's stores have reached the L2 (s_waitcnt vmcnt(0): the L1 writes through)
      __syncthreads();
      if (tid == 0) st_i32(&ka->D.segDone[u], seg + 1);
      continue;
    }
    if (status != DSR_OK) {
      // abort: the tagged state table needs no cleaning
      if (tid == 0) {
        dsr_decode_result* const res = ka->res; clear_result(&res[u]); res[u].status = status; res[u].frames = T - 1;
        if (segS > 0) { st_i32(&ka->D.segState[u].status, status); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_i32(&ka->D.segDone[u], 0x7FFFFFFF); }   // later segments pass by
      }
    }
  }
  RELOAD();
  if (tid == 0) ka->D.tags[slot] = s_tag;
  if (PROF && tid < 32) ka->D.prof[slot * 32 + tid] = s_prof[tid];
#undef TICK
#undef RELOAD
#undef TOKA
#undef TOKB
#undef curA
#undef nxtA
#undef curB
#undef nxtB
#undef sprA
#undef sprB
#undef ctok
#undef side
#undef cA
#undef cB
#undef arena
#undef sc
#undef dump
}

struct DecoderState {
  dsr_decoder_cfg cfg; bool haveGraph = false; int nSlots = 0; int nNodes = 0;
  WfstGraph::Csr csr; WfstGraph::Tables tab;
  DevBuf<int> d_xoff, d_xarc, d_xpathOff, d_eoff, d_path, d_nodeFinal, d_queue; DevBuf<float> d_pathCost;
  DevBuf<XRec> d_xrec; DevBuf<ERec> d_erec; DevBuf<float> d_arcCost, d_nodeCost; DevBuf<uint32_t> d_arcOut, d_arcIn;
  DevBuf<TokA> d_tokA, d_ctok; DevBuf<TokB> d_tokB; DevBuf<Side> d_side; DevBuf<XRecD> d_xrecD; int fastOK = 0; int maxCnt = 0; DevBuf<int> d_tokOff, d_tokCnt, d_owner, d_rank, d_chead; DevBuf<unsigned> d_tags; DevBuf<CandA> d_cA; DevBuf<CandB> d_cB; DevBuf<unsigned> d_first; DevBuf<Bp> d_arena;
  DevBuf<long long> d_prof; DevBuf<dsr_decode_result> d_res; DevBuf<int> d_arcs; DevBuf<unsigned> d_words;
  long arenaCap = 0; int initial = 0; unsigned tokenMemoryLimit = 0;
  long lastPoolCap = 0;
  DevBuf<SegState> d_segState; DevBuf<int> d_segDone; DevBuf<unsigned long long> d_poolNext; DevBuf<TokA> d_saveA; DevBuf<TokB> d_saveB;   // time slicing (DecDev::segFrames)
  // DecoderWordTrace mode (cfg.wordTrace): scratch of k_wordtrace.hip
  DevBuf<WTok> w_tok; DevBuf<WCand> w_cand; DevBuf<int> w_tokOff, w_rank; DevBuf<unsigned long long> w_best; DevBuf<unsigned> w_first; DevBuf<int4> w_traces; size_t w_tablesFor = 0;
  bool costNegZero = false; double costMinAbs = HUGE_VAL;      // over the arcs of the transducer set last: a cost of -0.0; the smallest non-zero |cost|
  // lattice bookkeeping of the last decode (cfg.latticeTokens > 0), per utterance
  DevBuf<uint4> d_lat; DevBuf<double> d_latTtl; DevBuf<long> d_latFrameOff; DevBuf<int> d_arenaLat; DevBuf<int4> d_latFinal; DevBuf<int> d_latInfo;
  int latU = 0, latTmax = 0; long latArenaCap = 0; WfstGraph graphCopy; DevBuf<TokA> d_tokA3; DevBuf<TokB> d_tokB3;
  // symbol tables of the transducer set last (borrowed) and the resolved silSymbol / eosSymbol (decoder.h:740-745)
  const dsr_lexicon* lexIn = nullptr; const dsr_lexicon* lexOut = nullptr; uint32_t eosX = 0; std::string eosSymbol;
  // what the last collected decode left: per-utterance results and best paths (pinned staging memory), for bestHypo / bestPath / finalStatesN
  int lastU = 0; size_t lastMaxPath = 0; bool lastPaths = false;
  PinBuf<dsr_decode_result> h_res; PinBuf<int> h_arcs; PinBuf<unsigned> h_words; hipEvent_t evDone = nullptr;
  int pendingU = 0; size_t pendingPath = 0; int pendingSlots = 0; long long* pendingProf = nullptr;
  // dump
  int dumpOn = 0; long dumpCap = 0; DevBuf<long> d_dumpFrameOff, d_dumpCount; DevBuf<int> d_dumpNode, d_dumpArc; DevBuf<float> d_dumpAc, d_dumpLm;
  std::vector<int64_t> h_dumpFrameOff; std::vector<int32_t> h_dumpNode, h_dumpArc; std::vector<float> h_dumpAc, h_dumpLm; int64_t h_dumpFrames = 0;
};

}  // namespace dsr

using namespace dsr;
struct dsr_wfst : WfstGraph { dsr_lexicon* lexState = nullptr; dsr_lexicon* lexIn = nullptr; dsr_lexicon* lexOut = nullptr; };
struct dsr_decoder : DecoderState {};

extern "C" {

dsr_status dsr_wfst_create(dsr_wfst** out) { return guard([&] { if (!out) throw Error(DSR_E_PARAMETER, "null argument"); *out = new dsr_wfst(); }); }
// WFSTFlyWeightSortedOutput(statelex, inlex, outlex) (decoder.i; wfstFlyWeight.h:403-424): the same container with every node's arcs kept ordered by
// (output, input); call on an empty transducer
dsr_status dsr_wfst_set_sorted_output(dsr_wfst* g, int on)
{ return guard([&] { if (!g) throw Error(DSR_E_PARAMETER, "null argument"); if (!g->arcs.empty()) throw Error(DSR_E_CONSISTENCY, "the transducer already has arcs"); g->sortedOutput = on != 0; }); }
void dsr_wfst_destroy(dsr_wfst* g) { delete g; }
dsr_status dsr_wfst_read(dsr_wfst* g, const char* f, int binary) { return guard([&] { if (!g) throw Error(DSR_E_PARAMETER, "null argument"); g->read(f, binary != 0); }); }
dsr_status dsr_wfst_read_dynamic(dsr_wfst* g, const char* f, int noSelfLoops) { return guard([&] { if (!g) throw Error(DSR_E_PARAMETER, "null argument"); g->readEx(f, false, noSelfLoops != 0); }); }
dsr_status dsr_wfst_write(const dsr_wfst* g, const char* f, int binary) { return guard([&] { if (!g) throw Error(DSR_E_PARAMETER, "null argument"); g->write(f, binary != 0); }); }
// WFSTFlyWeight::write(fileName, binary, useSymbols) (wfstFlyWeight.cc:415-463): with useSymbols every arc line carries the lexica's strings (Edge::write
// :499-516: states too when the state lexicon is non-empty, costs below 1e-4 left out); final-state lines and -- with binary -- the end marker stay numeric
dsr_status dsr_wfst_write_symbols(const dsr_wfst* g, const char* f, int binary, int useSymbols)
{ return guard([&] { if (!g) throw Error(DSR_E_PARAMETER, "null argument"); g->write(f, binary != 0, useSymbols != 0); }); }
// WFSTFlyWeight::reverse(wfst) (:141-213) and reverseRead(fileName) (:215-297)
dsr_status dsr_wfst_reverse(dsr_wfst* g, const dsr_wfst* src)
{ return guard([&] { if (!g || !src) throw Error(DSR_E_PARAMETER, "null argument"); g->reverse(*src); }); }
dsr_status dsr_wfst_reverse_read(dsr_wfst* g, const char* f)
{ return guard([&] { if (!g) throw Error(DSR_E_PARAMETER, "null argument"); g->reverseRead(f); }); }
dsr_status dsr_wfst_add_arc(dsr_wfst* g, unsigned s1, unsigned s2, unsigned in, unsigned out, float cost)
{ return guard([&] { if (!g) throw Error(DSR_E_PARAMETER, "null argument"); g->addArc(s1, s2, in, out, cost, true); }); }
dsr_status dsr_wfst_add_final(dsr_wfst* g, unsigned s, float cost) { return guard([&] { if (!g) throw Error(DSR_E_PARAMETER, "null argument"); g->addFinal(s, cost); }); }
int dsr_wfst_num_nodes(const dsr_wfst* g) { return (int) g->nodes.size(); }
int dsr_wfst_num_arcs(const dsr_wfst* g) { return (int) g->arcs.size(); }
dsr_status dsr_wfst_export(const dsr_wfst* g, uint32_t* nodeState, int32_t* nodeFinal, float* nodeCost, int32_t* arcOff,
                           int32_t* arcDst, uint32_t* arcIn, uint32_t* arcOut, float* arcCost)
{
  return guard([&] {
    if (!g) throw Error(DSR_E_PARAMETER, "null argument");
    const WfstGraph::Csr c = g->csr(); const size_t n = g->nodes.size();
    for (size_t i = 0; i < n; i++) { if (nodeState) nodeState[i] = g->nodes[i].state; if (nodeFinal) nodeFinal[i] = g->nodes[i].final_; if (nodeCost) nodeCost[i] = g->nodes[i].cost; }
    if (arcOff) for (size_t i = 0; i <= n; i++) arcOff[i] = c.off[i];
    for (size_t a = 0; a < c.dst.size(); a++) { if (arcDst) arcDst[a] = c.dst[a]; if (arcIn) arcIn[a] = c.in[a]; if (arcOut) arcOut[a] = c.out[a]; if (arcCost) arcCost[a] = c.cost[a]; }
  });
}

// WFSTFlyWeight(statelex, inlex, outlex) (decoder.i:52-70): the lexica are borrowed (the reference holds reference-counted pointers); the text reader
// looks non-numeric fields up in them (wfstFlyWeight.cc:311-347)
dsr_status dsr_wfst_set_lexicons(dsr_wfst* g, dsr_lexicon* stateLex, dsr_lexicon* inputLex, dsr_lexicon* outputLex)
{
  return guard([&] {
    if (!g) throw Error(DSR_E_PARAMETER, "null argument");
    g->lexState = stateLex; g->lexIn = inputLex; g->lexOut = outputLex;
    g->symbolOf = [g](int which, const char* t) -> uint32_t {
      dsr_lexicon* l = which == 0 ? g->lexState : which == 1 ? g->lexIn : g->lexOut;
      if (!l) throw Error(DSR_E_KEY, "field '%s' is not a number and the transducer has no %s lexicon", t, which == 0 ? "state" : which == 1 ? "input" : "output");
      return l->index(t);
    };
    g->nameOf = [g](int which, uint32_t i) -> std::string {
      const dsr_lexicon* l = which == 0 ? g->lexState : which == 1 ? g->lexIn : g->lexOut;
      if (!l) throw Error(DSR_E_KEY, "the transducer has no %s lexicon", which == 0 ? "state" : which == 1 ? "input" : "output");
      return l->symbol(i);
    };
    g->stateLexSize = [g]() -> size_t { return g->lexState ? g->lexState->syms.size() : 0; };
  });
}
dsr_lexicon* dsr_wfst_state_lexicon(const dsr_wfst* g) { return g ? g->lexState : nullptr; }
dsr_lexicon* dsr_wfst_input_lexicon(const dsr_wfst* g) { return g ? g->lexIn : nullptr; }
dsr_lexicon* dsr_wfst_output_lexicon(const dsr_wfst* g) { return g ? g->lexOut : nullptr; }
int dsr_wfst_has_final_state(const dsr_wfst* g) { if (!g) return 0; for (size_t i = 0; i < g->nodes.size(); i++) if (g->nodes[i].final_) return 1; return 0; }

void dsr_decoder_default_cfg(dsr_decoder_cfg* c)
{ memset(c, 0, sizeof(*c)); c->beam = 100.0; c->lmScale = 12.0; c->lmPenalty = 0.0; c->silPenalty = 0.0; c->silenceX = 0xFFFFFFFFu;
  c->propagateN = 5; c->wordTraceLattice = 1; }                            // DecoderWordTrace's defaults (decoder.i:201-260; only read when wordTrace != 0)

dsr_status dsr_decoder_create(const dsr_decoder_cfg* cfg, dsr_decoder** out)
{
  return guard([&] {
    if (!cfg || !out) throw Error(DSR_E_PARAMETER, "null argument");
    require_device();
    dsr_decoder* d = new dsr_decoder(); d->cfg = *cfg;
    if (d->cfg.maxActive <= 0) d->cfg.maxActive = 65536;
    if (d->cfg.maxCandidates <= 0) d->cfg.maxCandidates = 8 * d->cfg.maxActive;
    if (d->cfg.maxCandidates >= (1 << 24)) throw Error(DSR_E_PARAMETER, "maxCandidates must be < 2^24");
    if (d->cfg.streams <= 0) {
      hipDeviceProp_t prop; int dev = 0; DSR_HIP(hipGetDevice(&dev)); DSR_HIP(hipGetDeviceProperties(&prop, dev));
      d->cfg.streams = prop.multiProcessorCount;
      if (const char* e = getenv("DSR_VITERBI_SLOTS")) { const int t = atoi(e); if (t > 0) d->cfg.streams = t; }
    }
    *out = d;
  });
}
void dsr_decoder_destroy(dsr_decoder* d) { if (d && d->evDone) (void) hipEventDestroy(d->evDone); delete d; }

dsr_status dsr_decoder_set(dsr_decoder* d, const dsr_wfst* g)
{
  return guard([&] {
    if (!d || !g) throw Error(DSR_E_PARAMETER, "null argument");
    if (g->initial < 0) throw Error(DSR_E_CONSISTENCY, "the transducer has no arcs");
    d->csr = g->csr(); d->tab = g->tables(d->csr, (size_t) 1 << 28);
    d->nNodes = (int) g->nodes.size(); d->initial = g->initial; d->graphCopy = *g;
    std::vector<int> nf(d->nNodes); std::vector<float> nc(d->nNodes);
    for (int i = 0; i < d->nNodes; i++) { nf[i] = g->nodes[i].final_; nc[i] = g->nodes[i].cost; }
    d->d_xoff.upload(d->tab.xoff); d->d_eoff.upload(d->tab.eoff); d->d_path.upload(d->tab.path);
    if (d->tab.xrec.empty()) throw Error(DSR_E_CONSISTENCY, "the transducer has no emitting arcs");
    d->d_xrec.upload(d->tab.xrec); d->d_xarc.upload(d->tab.xarc); d->d_xpathOff.upload(d->tab.xpathOff);
    {
      // expansion records with the destination's own expansion range folded in (the register path never reads xoff)
      const size_t nx = d->tab.xrec.size(); std::vector<XRecD> xd(nx); int maxCnt = 0; std::vector<float> pc(1, 0.0f);
      for (size_t r = 0; r < nx; r++) {
        const XRec& x = d->tab.xrec[r]; XRecD& o = xd[r];
        o.dst = x.dst; o.dist = x.dist; o.cost = x.cost; o.meta = x.meta; o.dstXoff = d->tab.xoff[x.dst]; o.dstCnt = d->tab.xoff[x.dst + 1] - d->tab.xoff[x.dst];
        const int po = d->tab.xpathOff[r], plen = (int) (x.meta & 0xFFFFu); o.p2 = 0; o.eps1 = 0.0f;
        if (x.meta >> 18) throw Error(DSR_E_CONSISTENCY, "expansion record %zu: meta bits above 17 are in use", r);
        if (plen) { const int a0 = d->tab.path[po]; o.eps1 = d->csr.cost[a0]; if (d->csr.out[a0] != 0) o.meta |= 0x20000u; }
        if (plen > 14) o.meta |= 0x80000000u;
        else if (plen > 1) {
          if (plen == 2) { const float c1 = d->csr.cost[d->tab.path[po + 1]]; memcpy(&o.p2, &c1, 4); }
          else { o.p2 = (int) pc.size(); for (int h = 1; h < plen; h++) pc.push_back(d->csr.cost[d->tab.path[po + h]]); }
          for (int h = 1; h < plen; h++) if (d->csr.out[d->tab.path[po + h]] != 0) o.meta |= 1u << (17 + h);
        }
        if (o.dstCnt > maxCnt) maxCnt = o.dstCnt;
      }
      d->d_xrecD.upload(xd); d->d_pathCost.upload(pc);
      d->fastOK = (maxCnt < (1 << 19) && nx < ((size_t) 1 << 27)) ? 1 : 0; d->maxCnt = maxCnt;       // (2^27 records x 32 bytes: the register path addresses them with 32-bit byte offsets)
      if (getenv("DSR_VITERBI_NOFAST")) d->fastOK = 0;
    }
    { std::vector<ERec> e = d->tab.erec; if (e.empty()) e.push_back(ERec{0, 0, 0, 0}); d->d_erec.upload(e); }
    d->d_arcCost.upload(d->csr.cost); d->d_arcOut.upload(d->csr.out); d->d_arcIn.upload(d->csr.in);
    d->costNegZero = false; d->costMinAbs = HUGE_VAL;
    for (size_t a = 0; a < d->csr.cost.size(); a++) {
      const float c = d->csr.cost[a]; uint32_t bits; memcpy(&bits, &c, 4);
      if (bits == 0x80000000u) d->costNegZero = true;
      if (c != 0.0f && std::fabs((double) c) < d->costMinAbs) d->costMinAbs = std::fabs((double) c);
    }
    d->d_nodeFinal.upload(nf); d->d_nodeCost.upload(nc);
    d->haveGraph = true; d->nSlots = 0;    // scratch is (re)allocated by the first decode
  });
}
// DecoderFlyWeight::set(wfst) = _Decoder::_set (decoder.h:740-745): the network and, through its lexica, the indices of silSymbol (input
// lexicon) and eosSymbol (output lexicon); a missing symbol is the reference's jkey_error (mlist.h:109-114).  Symbols may be NULL when the
// transducer carries no lexica (then cfg.silenceX stays as configured).
dsr_status dsr_decoder_set_symbols(dsr_decoder* d, const dsr_wfst* g, const char* silSymbol, const char* eosSymbol)
{
  return guard([&] {
    if (!d || !g) throw Error(DSR_E_PARAMETER, "null argument");
    uint32_t silX = d->cfg.silenceX, eosX = 0;
    if (silSymbol) { if (!g->lexIn) throw Error(DSR_E_KEY, "the transducer has no input lexicon to look '%s' up in", silSymbol); silX = g->lexIn->index(silSymbol); }
    if (eosSymbol) { if (!g->lexOut) throw Error(DSR_E_KEY, "the transducer has no output lexicon to look '%s' up in", eosSymbol); eosX = g->lexOut->index(eosSymbol); }
    const dsr_status s = dsr_decoder_set(d, g);
    if (s != DSR_OK) throw Error(s, "%s", dsr_last_error());
    d->cfg.silenceX = silX; d->eosX = eosX; d->lexIn = g->lexIn; d->lexOut = g->lexOut; d->eosSymbol = eosSymbol ? eosSymbol : "";
  });
}
uint32_t dsr_decoder_eos_index(const dsr_decoder* d) { return d ? d->eosX : 0; }

// Results of the last collected decode, utterance u.  bestHypo(useInputSymbols) (decoder.h:748-773): the output symbols != 0 along the best path, or
// the input symbols != 0 with immediate repetitions dropped ("inX != 0 && inX != lastX", walking the path from its END, so a repetition is judged
// against the symbol after it), each followed by a blank.  bestPath() (:775-797): the names of the distributions along the path = the input
// symbols != 0, one per line.  Strings need the lexica (DSR_E_KEY without); the id variants do not.
static void need_last(const dsr_decoder* d, int u, bool paths)
{
  if (!d) throw Error(DSR_E_PARAMETER, "null argument");
  if (d->lastU <= 0) throw Error(DSR_E_CONSISTENCY, "no decode has been collected yet");
  if (u < 0 || u >= d->lastU) throw Error(DSR_E_INDEX, "utterance %d of %d", u, d->lastU);
  if (paths && !d->lastPaths) throw Error(DSR_E_CONSISTENCY, "the last decode was collected without its paths");
  if (d->h_res.p[u].status != DSR_OK) throw Error(d->h_res.p[u].status, "utterance %d was not decoded (status %d)", u, d->h_res.p[u].status);
}
// ids: the path's symbol ids in time order (which: 0 outputs != 0; 1 inputs != 0 with repetitions dropped as bestHypo(true); 2 inputs != 0 as bestPath)
static std::vector<uint32_t> path_ids(const dsr_decoder* d, int u, int which)
{
  need_last(d, u, true);
  const dsr_decode_result& r = d->h_res.p[u];
  const int n = r.nArcs < (int) d->lastMaxPath ? r.nArcs : (int) d->lastMaxPath;
  const int* arcs = d->h_arcs.p + (size_t) u * d->lastMaxPath;
  std::vector<uint32_t> ids;
  if (which == 1) {                                              // from the end, as the reference walks prev(): keep inX when it differs from the LAST KEPT one
    uint32_t lastX = 0;
    for (int i = n - 1; i >= 0; i--) { const uint32_t inX = d->csr.in[arcs[i]]; if (inX != 0 && inX != lastX) { ids.push_back(inX); lastX = inX; } }
    std::reverse(ids.begin(), ids.end());
  } else for (int i = 0; i < n; i++) { const uint32_t v = which == 0 ? d->csr.out[arcs[i]] : d->csr.in[arcs[i]]; if (v != 0) ids.push_back(v); }
  return ids;
}
dsr_status dsr_decoder_path_ids(const dsr_decoder* d, int u, int which, uint32_t* ids, int cap, int* n)
{
  return guard([&] {
    if (!n || which < 0 || which > 2) throw Error(DSR_E_PARAMETER, "bad argument");
    const std::vector<uint32_t> v = path_ids(d, u, which);
    *n = (int) v.size();
    if (ids) { if ((int) v.size() > cap) throw Error(DSR_E_DIMENSION, "buffer holds %d ids, the path has %zu", cap, v.size()); if (!v.empty()) memcpy(ids, v.data(), 4 * v.size()); }
  });
}
static void put_string(const std::string& s, char* buf, size_t cap, size_t* need)
{
  if (need) *need = s.size() + 1;
  if (buf) { if (s.size() + 1 > cap) throw Error(DSR_E_DIMENSION, "buffer holds %zu bytes, the string needs %zu", cap, s.size() + 1); memcpy(buf, s.c_str(), s.size() + 1); }
}
dsr_status dsr_decoder_best_hypo(const dsr_decoder* d, int u, int useInputSymbols, char* buf, size_t cap, size_t* need)
{
  return guard([&] {
    const std::vector<uint32_t> v = path_ids(d, u, useInputSymbols ? 1 : 0);
    const dsr_lexicon* lex = useInputSymbols ? d->lexIn : d->lexOut;
    if (!lex) throw Error(DSR_E_KEY, "the transducer set on this decoder has no %s lexicon", useInputSymbols ? "input" : "output");
    std::string s; for (size_t i = 0; i < v.size(); i++) { s += lex->symbol(v[i]); s += " "; }
    put_string(s, buf, cap, need);
  });
}
dsr_status dsr_decoder_best_path(const dsr_decoder* d, int u, char* buf, size_t cap, size_t* need, int* count)
{
  return guard([&] {
    const std::vector<uint32_t> v = path_ids(d, u, 2);
    if (!d->lexIn) throw Error(DSR_E_KEY, "the transducer set on this decoder has no input lexicon");
    std::string s; for (size_t i = 0; i < v.size(); i++) { s += d->lexIn->symbol(v[i]); s += "\n"; }
    if (count) *count = (int) v.size();
    put_string(s, buf, cap, need);
  });
}
dsr_status dsr_decoder_final_states_n(const dsr_decoder* d, int u, int* n)
{ return guard([&] { if (!n) throw Error(DSR_E_PARAMETER, "null argument"); need_last(d, u, false); *n = d->h_res.p[u].finalStatesN; }); }
dsr_status dsr_decoder_trace_back_succeeded(const dsr_decoder* d, int u, int* ok)
{ return guard([&] { if (!ok) throw Error(DSR_E_PARAMETER, "null argument"); need_last(d, u, false); *ok = d->h_res.p[u].reachedFinal; }); }

// _Decoder::setTokenMemoryLimit(limit) (decoder.h:396) caps the reference's Token memory pool (MemoryManager).  Tokens here live in per-slot
// arrays sized by cfg.maxActive / maxCandidates / arenaTokens (a decode that outgrows them returns DSR_E_ALLOCATION for that utterance): there
// is no pool to limit.  The value is accepted and kept so that drivers that set it run unchanged.
dsr_status dsr_decoder_set_token_memory_limit(dsr_decoder* d, unsigned limit)
{ return guard([&] { if (!d) throw Error(DSR_E_PARAMETER, "null argument"); d->tokenMemoryLimit = limit; }); }
unsigned dsr_decoder_token_memory_limit(const dsr_decoder* d) { return d ? d->tokenMemoryLimit : 0u; }

dsr_status dsr_decoder_set_beam(dsr_decoder* d, double beam) { return guard([&] { if (!d) throw Error(DSR_E_PARAMETER, "null argument"); d->cfg.beam = beam; }); }

dsr_status dsr_decoder_enable_dump(dsr_decoder* d, int en) { return guard([&] { if (!d) throw Error(DSR_E_PARAMETER, "null argument"); d->dumpOn = en; }); }

// Which XCD a workgroup lands on: HW_REG_XCC_ID of workgroup i of a 64-workgroup grid.  The XCD-bound queues of the time-sliced decode rest on two properties of the
// dispatcher that are checked here once per process instead of assumed: eight XCDs numbered 0..7, and workgroup i of a grid on XCD (i mod 8) -- so that a grid of 8 k
// workgroups serves every queue.
__global__ void k_xcc_probe(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = (int) (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u); }
static bool xcd_round_robin(hipStream_t st)
{
  static int known = -1;
  if (known >= 0) return known == 1;
  DevBuf<int> o; o.reserve(64); int h[64];
  hipLaunchKernelGGL(k_xcc_probe, dim3(64), dim3(64), 0, st, o.p);
  DSR_HIP(hipMemcpyAsync(h, o.p, sizeof(h), hipMemcpyDeviceToHost, st)); DSR_HIP(hipStreamSynchronize(st));
  bool ok = true; unsigned seen = 0;
  for (int i = 0; i < 64; i++) { if (h[i] < 0 || h[i] > 7 || h[i] != h[i & 7]) ok = false; }
  for (int i = 0; i < 8 && ok; i++) seen |= 1u << h[i];
  known = (ok && seen == 0xFFu) ? 1 : 0;
  return known == 1;
}

static void ensure_scratch(dsr_decoder* d, int slots, int Tmax, int arenas)
{
  const dsr_decoder_cfg& c = d->cfg;
  long arena = c.arenaTokens > 0 ? (long) c.arenaTokens : (long) 8192 * (long) (Tmax + 2);
  if (arena > 0x7FFFFFF0L) arena = 0x7FFFFFF0L;
  if (slots <= d->nSlots && arena <= d->arenaCap && (size_t) (arenas > slots ? arenas : slots) * (size_t) arena <= d->d_arena.n) return;
  if (slots < d->nSlots) slots = d->nSlots;
  if (arena < d->arenaCap) arena = d->arenaCap;
  const size_t S = (size_t) slots;
  d->d_tokA.reserve(S * 2 * c.maxActive); d->d_tokB.reserve(S * 2 * c.maxActive); d->d_ctok.reserve(S * 8192); d->d_side.reserve(S * kFastC);
  d->d_tokOff.reserve(S * (c.maxActive + 1));
  d->d_owner.reserve(S * c.maxCandidates); d->d_rank.reserve(S * c.maxCandidates);
  d->d_cA.reserve(S * c.maxCandidates); d->d_cB.reserve(S * c.maxCandidates);
  d->d_first.reserve(S * d->nNodes); d->d_tokCnt.reserve(S * (c.maxActive + 1)); d->d_chead.reserve(S * c.maxCandidates);
  d->d_tags.reserve(S); DSR_HIP(hipMemset(d->d_tags.p, 0, S * sizeof(unsigned)));          // tag 0 = wipe the table on first use
  d->d_arena.reserve((size_t) (arenas > slots ? arenas : slots) * (size_t) arena);
  d->d_queue.reserve(8);
  d->nSlots = slots; d->arenaCap = arena;
}

// Asynchronous halves of decode_batch: launch enqueues the kernel and the copies of the results into pinned staging
// memory on `stream` and returns; collect waits for that work and hands the results out.  One launch may be in flight
// per decoder object (its scratch memory belongs to the launch).
dsr_status dsr_decoder_decode_launch(dsr_decoder* d, const float* score, const int32_t* nframes, int U, int Tmax, int nDist,
                                     int maxPath, int want_paths, void* stream)
{
  return guard([&] {
    if (!d || !score || !nframes) throw Error(DSR_E_PARAMETER, "null argument");
    if (!d->haveGraph) throw Error(DSR_E_INITIALIZATION, "call set() with a transducer first");
    if (d->pendingU > 0) throw Error(DSR_E_CONSISTENCY, "a decode is already in flight on this decoder: collect it first");
    if (U <= 0) return;
    hipStream_t st = (hipStream_t) stream;
    if (d->cfg.wordTrace) {
      // DecoderWordTrace (decoder.h:1146-1304).  With generateLattice -- the reference's default -- _placeOnList merges the worse chains of two tokens and reads
      // wordTrace()->wordSequenceX() of each (decoder.cc:239); a token that has not crossed a word boundary has a null word trace: undefined behaviour in the
      // reference on any transducer whose first arcs carry no output symbol.  That search is not built; the 1-best search (generateLattice = false) is.
      if (d->cfg.wordTraceLattice) throw Error(DSR_E_CONSISTENCY, "DecoderWordTrace with generateLattice: not built (the shipped _placeOnList dereferences the null word trace of "
                                               "every token that has not crossed a word boundary, decoder.cc:239); construct it with generateLattice = false");
      if (d->cfg.topN > 0 || d->cfg.latticeTokens > 0 || d->dumpOn) throw Error(DSR_E_PARAMETER, "DecoderWordTrace: no topN (its frame loop has no such branch, decoder.cc:147-183), lattice bookkeeping or dump");
      for (size_t a = 0; a < d->csr.in.size(); a++) if (d->csr.in[a] > (uint32_t) nDist) throw Error(DSR_E_INDEX, "arc input %u has no distribution (nDist=%d)", d->csr.in[a], nDist);
      int slots = d->cfg.streams; if (slots > U) slots = U;
      const dsr_decoder_cfg& c = d->cfg; const size_t S = (size_t) slots;
      const long maxTraces = c.wordTraces > 0 ? (long) c.wordTraces : (long) 1 << 20;
      d->w_tok.reserve(S * 2 * c.maxActive); d->w_cand.reserve(S * c.maxCandidates); d->w_tokOff.reserve(S * (c.maxActive + 1)); d->w_rank.reserve(S * c.maxCandidates);
      d->w_traces.reserve((size_t) U * maxTraces); d->d_queue.reserve(8); d->d_res.reserve(U);
      if (d->w_tablesFor != S * d->nNodes) {                               // the per-state tables are all ones between frames: set once, the kernel restores them
        d->w_best.reserve(S * d->nNodes); d->w_first.reserve(S * d->nNodes); d->w_tablesFor = S * d->nNodes;
        DSR_HIP(hipMemsetAsync(d->w_best.p, 0xFF, sizeof(unsigned long long) * S * d->nNodes, st)); DSR_HIP(hipMemsetAsync(d->w_first.p, 0xFF, sizeof(unsigned) * S * d->nNodes, st));
      }
      if (maxPath < 1) maxPath = 1;
      d->d_arcs.reserve((size_t) U * maxPath); d->d_words.reserve((size_t) U * maxPath);
      DSR_HIP(hipMemsetAsync(d->d_queue.p, 0, 8 * sizeof(int), st));
      WtArgs A; A.nNodes = d->nNodes; A.initial = d->initial; A.xoff = d->d_xoff.p; A.xrec = d->d_xrec.p; A.xarc = d->d_xarc.p; A.xpathOff = d->d_xpathOff.p; A.eoff = d->d_eoff.p;
      A.erec = d->d_erec.p; A.path = d->d_path.p; A.arcCost = d->d_arcCost.p; A.arcOut = d->d_arcOut.p; A.arcIn = d->d_arcIn.p; A.nodeFinal = d->d_nodeFinal.p; A.nodeCost = d->d_nodeCost.p;
      A.beam = c.beam; A.lmScale = c.lmScale; A.lmPenalty = c.lmPenalty; A.silPenalty = c.silPenalty; A.silenceX = c.silenceX; A.insertSilence = c.insertSilence;
      A.maxTok = c.maxActive; A.maxCand = c.maxCandidates; A.maxTraces = maxTraces;
      A.tok = d->w_tok.p; A.cand = d->w_cand.p; A.tokOff = d->w_tokOff.p; A.rank = d->w_rank.p; A.bestKey = d->w_best.p; A.firstSlot = d->w_first.p; A.traces = d->w_traces.p; A.queue = d->d_queue.p;
      A.scores = score; A.nframes = nframes; A.U = U; A.Tmax = Tmax; A.nDist = nDist; A.res = d->d_res.p; A.arcsOut = d->d_arcs.p; A.wordsOut = d->d_words.p; A.maxPath = maxPath;
      wordtrace_launch(A, slots, st);
      const size_t nPath = want_paths ? (size_t) U * maxPath : 0;
      d->h_res.reserve(U); d->h_arcs.reserve(nPath ? nPath : 1); d->h_words.reserve(nPath ? nPath : 1);
      DSR_HIP(hipMemcpyAsync(d->h_res.p, d->d_res.p, sizeof(dsr_decode_result) * U, hipMemcpyDeviceToHost, st));
      if (nPath) DSR_HIP(hipMemcpyAsync(d->h_arcs.p, d->d_arcs.p, sizeof(int) * nPath, hipMemcpyDeviceToHost, st));
      if (nPath) DSR_HIP(hipMemcpyAsync(d->h_words.p, d->d_words.p, sizeof(unsigned) * nPath, hipMemcpyDeviceToHost, st));
      if (!d->evDone) DSR_HIP(hipEventCreateWithFlags(&d->evDone, hipEventDisableTiming));
      DSR_HIP(hipEventRecord(d->evDone, st));
      d->pendingU = U; d->pendingPath = nPath; d->pendingSlots = slots; d->pendingProf = nullptr; d->latU = 0;
      return;
    }
    int32_t* arcs_out = want_paths ? (int32_t*) 1 : nullptr; uint32_t* words_out = want_paths ? (uint32_t*) 1 : nullptr;     // (only tested for null below)
    // every input symbol must name a distribution (decoder.h:985: _dist->find(distX-1))
    for (size_t a = 0; a < d->csr.in.size(); a++) if (d->csr.in[a] > (uint32_t) nDist) throw Error(DSR_E_INDEX, "arc input %u has no distribution (nDist=%d)", d->csr.in[a], nDist);
    int slots = d->cfg.streams; if (slots > U) slots = U; if (d->dumpOn) slots = 1;
    const bool latOn = d->cfg.latticeTokens > 0;
    if (latOn && d->dumpOn) throw Error(DSR_E_PARAMETER, "lattice bookkeeping and the token dump are separate debugging aids: enable one");
    // Time slicing (DecDev): when there are more utterances than workgroups, in the plain decode mode.  DSR_VITERBI_SEG = frames per segment (0: run every utterance to completion).
    int segFrames = 0;
    if (!latOn && !d->dumpOn && d->cfg.topN <= 0 && U > slots) {
      segFrames = getenv("DSR_VITERBI_SEG") ? atoi(getenv("DSR_VITERBI_SEG")) : 125;
      if (segFrames < 0 || 2 * segFrames > Tmax + 1) segFrames = 0;
      // between its segments an utterance's token list waits in a save area of maxActive tokens: with very large lists and very many utterances that is more memory
      // than the scheduling is worth (DSR_VITERBI_SEG_SAVE_GB, default 16)
      const double saveGB = (double) U * (double) d->cfg.maxActive * 24.0 / 1e9;
      if (saveGB > (getenv("DSR_VITERBI_SEG_SAVE_GB") ? atof(getenv("DSR_VITERBI_SEG_SAVE_GB")) : 16.0)) segFrames = 0;
    }
    // XCD-bound queues (the cheap hand-over) need workgroups on every XCD: grids of 8 k >= 64 workgroups on a device that deals workgroups out round robin.
    // Anything else decodes every utterance in one go -- unless DSR_VITERBI_SEG_ANY asks for the one-queue form (device-scope fences at every hand-over: the tests).
    int segQueues = 8;
    if (segFrames > 0 && !(slots >= 64 && slots % 8 == 0 && xcd_round_robin((hipStream_t) stream))) { if (getenv("DSR_VITERBI_SEG_ANY")) segQueues = 1; else segFrames = 0; }
    if (segFrames > 0 && getenv("DSR_VITERBI_SEG_ANY") && atoi(getenv("DSR_VITERBI_SEG_ANY")) == 2) segQueues = 1;
    // (sliced: one pool of back-pointer records for the batch instead of an arena per slot -- 1536 records per utterance and frame on average, at least what the slots had)
    const long arenaPer = d->cfg.arenaTokens > 0 ? (long) d->cfg.arenaTokens : (long) 8192 * (long) (Tmax + 2);
    int poolArenas = 0;
    if (segFrames > 0) {
      const double want = (double) U * 1536.0 * (double) (Tmax + 2) / (double) arenaPer;
      poolArenas = (int) std::min<double>(std::ceil(want), (double) (0xFFFFFFF0u / (unsigned long long) arenaPer));
      if ((unsigned long long) std::max(poolArenas, slots) * (unsigned long long) arenaPer > 0xFFFFFFF0ull) segFrames = 0;      // back pointers are 32-bit pool indices
    }
    ensure_scratch(d, slots, Tmax, latOn ? U : (segFrames > 0 ? poolArenas : 0));
    d->d_res.reserve(U);
    if (maxPath < 0) maxPath = 0;
    if (arcs_out || words_out) { d->d_arcs.reserve((size_t) U * (maxPath > 0 ? maxPath : 1)); d->d_words.reserve((size_t) U * (maxPath > 0 ? maxPath : 1)); }
    DSR_HIP(hipMemsetAsync(d->d_queue.p, 0, 8 * sizeof(int), st));
    if (d->dumpOn) {
      d->dumpCap = (long) d->cfg.maxActive * 64 < (long) 1 << 26 ? (long) 1 << 24 : (long) 1 << 26;
      d->d_dumpFrameOff.reserve(Tmax + 2); d->d_dumpCount.reserve(2);
      d->d_dumpNode.reserve(d->dumpCap); d->d_dumpArc.reserve(d->dumpCap); d->d_dumpAc.reserve(d->dumpCap); d->d_dumpLm.reserve(d->dumpCap);
      DSR_HIP(hipMemsetAsync(d->d_dumpCount.p, 0, 2 * sizeof(long), st));
    }
    GraphDev G; G.nNodes = d->nNodes; G.initial = d->initial; G.xoff = d->d_xoff.p; G.xrec = d->d_xrec.p; G.xrecD = d->d_xrecD.p; G.xarc = d->d_xarc.p;
    G.xpathOff = d->d_xpathOff.p; G.eoff = d->d_eoff.p; G.erec = d->d_erec.p; G.path = d->d_path.p; G.pathCost = d->d_pathCost.p; G.arcCost = d->d_arcCost.p;
    G.arcOut = d->d_arcOut.p; G.arcIn = d->d_arcIn.p; G.nodeFinal = d->d_nodeFinal.p; G.nodeCost = d->d_nodeCost.p;
    DecDev D; D.beam = d->cfg.beam; D.lmScale = d->cfg.lmScale; D.lmPenalty = d->cfg.lmPenalty; D.silPenalty = d->cfg.silPenalty;
    // penalty-free expansion (k_viterbi, expandR): both penalty products zero, and no placement's lm can be -0.0 -- lmScale > 0, no arc cost of -0.0, and every
    // non-zero lmScale x cost at least 2^-60 in magnitude (a sum float + double of that size is 0 or at least 2^-112: it cannot round to -0.0f)
    D.noPen = (d->cfg.lmScale > 0.0 && std::isfinite(d->cfg.lmScale) && d->cfg.lmScale * d->cfg.lmPenalty == 0.0 && d->cfg.lmScale * d->cfg.silPenalty == 0.0 &&
               !d->costNegZero && (d->costMinAbs == HUGE_VAL || d->cfg.lmScale * d->costMinAbs >= 0x1p-60) && !getenv("DSR_VITERBI_PEN")) ? 1 : 0;
    D.silenceX = d->cfg.silenceX; D.maxTok = d->cfg.maxActive; D.maxCand = d->cfg.maxCandidates; D.arenaCap = d->arenaCap;
    D.tokA = d->d_tokA.p; D.tokB = d->d_tokB.p; D.ctok = d->d_ctok.p; D.side = d->d_side.p; D.fastOK = d->fastOK; D.tokOff = d->d_tokOff.p; D.owner = d->d_owner.p; D.rank = d->d_rank.p; D.cA = d->d_cA.p; D.cB = d->d_cB.p;
    D.first = d->d_first.p; D.tags = d->d_tags.p; D.tokCnt = d->d_tokCnt.p; D.chead = d->d_chead.p; D.arena = d->d_arena.p; D.queue = d->d_queue.p;
    if (getenv("DSR_VITERBI_SEG_VERBOSE")) fprintf(stderr, "[dsr viterbi] %d utterances on %d workgroups: %s\n", U, slots, segFrames > 0 ? (segQueues == 8 ? "time-sliced, XCD-bound queues" : "time-sliced, one queue") : "run to completion");
    D.segDrop = getenv("DSR_VITERBI_SEG_DROP") ? (int) strtol(getenv("DSR_VITERBI_SEG_DROP"), nullptr, 0) : 0;
    D.segQueues = segQueues; D.segFrames = segFrames; D.segCount = segFrames > 0 ? (Tmax + segFrames) / segFrames : 1;            // segments cover frames 0 .. Tmax (the end expansion is "frame" T)
    D.poolCap = 0; D.poolChunk = 0; D.poolNext = nullptr; D.segState = nullptr; D.segDone = nullptr; D.saveA = nullptr; D.saveB = nullptr;
    if (segFrames > 0) {
      d->d_segState.reserve(U); d->d_segDone.reserve(U); d->d_poolNext.reserve(1);
      d->d_saveA.reserve((size_t) U * d->cfg.maxActive); d->d_saveB.reserve((size_t) U * d->cfg.maxActive);
      DSR_HIP(hipMemsetAsync(d->d_segDone.p, 0, sizeof(int) * (size_t) U, st)); DSR_HIP(hipMemsetAsync(d->d_poolNext.p, 0, sizeof(unsigned long long), st));
      D.poolCap = (long) ((size_t) std::max(poolArenas, slots) * (size_t) d->arenaCap); if (D.poolCap > (long) 0xFFFFFFF0L) D.poolCap = (long) 0xFFFFFFF0L;
      D.poolChunk = std::min<long>(262144, std::max<long>(4096, D.poolCap / (4 * (long) U)));      // (what an utterance leaves unused of its last run: at most a quarter of the pool in all)
      D.poolNext = d->d_poolNext.p; D.segState = d->d_segState.p; D.segDone = d->d_segDone.p; D.saveA = d->d_saveA.p; D.saveB = d->d_saveB.p;
    }
    d->lastPoolCap = D.poolCap;
    D.prof = nullptr;
    if (getenv("DSR_VITERBI_PROF")) { d->d_prof.reserve((size_t) slots * 32); D.prof = d->d_prof.p; }
    D.dumpOn = d->dumpOn; D.dumpCap = d->dumpCap; D.dumpFrameOff = d->d_dumpFrameOff.p; D.dumpNode = d->d_dumpNode.p; D.dumpAc = d->d_dumpAc.p;
    D.dumpLm = d->d_dumpLm.p; D.dumpArc = d->d_dumpArc.p; D.dumpCount = d->d_dumpCount.p;
    D.topN = d->cfg.topN > 0 ? d->cfg.topN : 0; D.tokA3 = nullptr; D.tokB3 = nullptr;
    if (D.topN > 0) { d->d_tokA3.reserve((size_t) slots * d->cfg.maxActive); d->d_tokB3.reserve((size_t) slots * d->cfg.maxActive); D.tokA3 = d->d_tokA3.p; D.tokB3 = d->d_tokB3.p; }
    D.latOn = latOn ? 1 : 0; D.latCap = 0; D.lat = nullptr; D.latTtl = nullptr; D.latFrameOff = nullptr; D.arenaLat = nullptr; D.latFinal = nullptr; D.latInfo = nullptr;
    if (latOn) {
      const size_t cap = (size_t) d->cfg.latticeTokens;
      d->d_lat.reserve((size_t) U * cap); d->d_latTtl.reserve((size_t) U * cap); d->d_latFrameOff.reserve((size_t) U * (Tmax + 3));
      d->d_arenaLat.reserve((size_t) U * (size_t) d->arenaCap); d->d_latFinal.reserve((size_t) U * d->cfg.maxActive); d->d_latInfo.reserve((size_t) 4 * U);
      DSR_HIP(hipMemsetAsync(d->d_latInfo.p, 0, sizeof(int) * 4 * (size_t) U, st));
      D.latCap = (long) cap; D.lat = d->d_lat.p; D.latTtl = d->d_latTtl.p; D.latFrameOff = d->d_latFrameOff.p; D.arenaLat = d->d_arenaLat.p; D.latFinal = d->d_latFinal.p; D.latInfo = d->d_latInfo.p;
      d->latU = U; d->latTmax = Tmax; d->latArenaCap = d->arenaCap;
    } else d->latU = 0;
    // LDS: [score row][state table: 2 x hashN words][slot offsets of the expanding tokens]; the row stays in global memory
    // when it would push the state table below the size the register path needs.  One 1024-thread workgroup per CU (16384 buckets).  The region
    // after the table holds the slot offsets (u16 per expanding token) and later the LDS side records.
    // instantiations: bit 0 per-phase ticks (DSR_VITERBI_PROF), bit 1 lattice bookkeeping / topN / token dump compiled in
    const int modes = (D.prof ? 1 : 0) | ((latOn || D.topN > 0 || d->dumpOn) ? 2 : 0);
    const void* kfn = modes == 0 ? (const void*) k_viterbi<0> : modes == 1 ? (const void*) k_viterbi<1> : modes == 2 ? (const void*) k_viterbi<2> : (const void*) k_viterbi<3>;
    hipFuncAttributes fattr; DSR_HIP(hipFuncGetAttributes(&fattr, kfn));
    const size_t eoffB = (size_t) kSideLds * sizeof(Side); const size_t ldsCap = (size_t) 160 * 1024 - fattr.sharedSizeBytes;      // what the kernel's static LDS leaves of a CU's 160 KB
    const int hashMax = 16384;
    int useLds = (size_t) nDist * sizeof(float) <= 64 * 1024;
    if (useLds && (size_t) ((nDist + 3) & ~3) * sizeof(float) + (size_t) hashMax * 8 + eoffB > ldsCap) useLds = 0;
    const size_t rowB = useLds ? (size_t) ((nDist + 3) & ~3) * sizeof(float) : 16;
    int hashN = hashMax; while (hashN > 0 && rowB + (size_t) hashN * 8 + eoffB > ldsCap) hashN >>= 1;
    if (getenv("DSR_VITERBI_NOHASH")) hashN = 0;
    // + the expansion counts of up to 2048 tokens (u16) when the budget and the graph's largest fan-out allow
    int cntCap = (hashN > 0 && d->maxCnt < 65536 && rowB + (size_t) hashN * 8 + eoffB + 4096 <= ldsCap) ? 2048 : 0;
    if (getenv("DSR_VITERBI_NOCNT")) cntCap = 0;
    const size_t lds = rowB + (size_t) hashN * 8 + eoffB + 2 * (size_t) cntCap;
    VitArgs A; A.G = G; A.D = D; A.scores = score; A.nframesArr = nframes; A.U = U; A.Tmax = Tmax; A.nDist = nDist; A.res = d->d_res.p;
    A.arcsOut = (arcs_out || words_out) ? d->d_arcs.p : nullptr; A.wordsOut = (arcs_out || words_out) ? d->d_words.p : nullptr; A.maxPath = maxPath;
    A.useLdsRow = useLds; A.hashN = hashN; A.regionB = (int) eoffB; A.cntCap = cntCap;
#define DSR_LAUNCH_V(MM) { DSR_HIP(hipFuncSetAttribute((const void*) k_viterbi<MM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
      hipLaunchKernelGGL(k_viterbi<MM>, dim3(slots), dim3(kThreads), lds, st, A); }
    switch (modes) { case 0: DSR_LAUNCH_V(0) break; case 1: DSR_LAUNCH_V(1) break; case 2: DSR_LAUNCH_V(2) break; default: DSR_LAUNCH_V(3) break; }
#undef DSR_LAUNCH_V
    DSR_HIP(hipGetLastError());
    const size_t nPath = want_paths ? (size_t) U * maxPath : 0;
    d->h_res.reserve(U); d->h_arcs.reserve(nPath ? nPath : 1); d->h_words.reserve(nPath ? nPath : 1);
    DSR_HIP(hipMemcpyAsync(d->h_res.p, d->d_res.p, sizeof(dsr_decode_result) * U, hipMemcpyDeviceToHost, st));
    if (nPath) DSR_HIP(hipMemcpyAsync(d->h_arcs.p, d->d_arcs.p, sizeof(int) * nPath, hipMemcpyDeviceToHost, st));
    if (nPath) DSR_HIP(hipMemcpyAsync(d->h_words.p, d->d_words.p, sizeof(unsigned) * nPath, hipMemcpyDeviceToHost, st));
    if (!d->evDone) DSR_HIP(hipEventCreateWithFlags(&d->evDone, hipEventDisableTiming));
    DSR_HIP(hipEventRecord(d->evDone, st));
    d->pendingU = U; d->pendingPath = nPath; d->pendingSlots = slots; d->pendingProf = D.prof;
  });
}

dsr_status dsr_decoder_decode_collect(dsr_decoder* d, dsr_decode_result* res, int32_t* arcs_out, uint32_t* words_out)
{
  return guard([&] {
    if (!d || !res) throw Error(DSR_E_PARAMETER, "null argument");
    if (d->pendingU <= 0) throw Error(DSR_E_CONSISTENCY, "no decode in flight");
    const int U = d->pendingU; const size_t nPath = d->pendingPath; const int slots = d->pendingSlots; long long* prof = d->pendingProf;
    d->pendingU = 0; d->lastU = U; d->lastPaths = nPath > 0; d->lastMaxPath = nPath > 0 ? nPath / (size_t) U : 0;
    DSR_HIP(hipEventSynchronize(d->evDone));
    memcpy(res, d->h_res.p, sizeof(dsr_decode_result) * U);
    if (arcs_out && nPath) memcpy(arcs_out, d->h_arcs.p, sizeof(int) * nPath);
    if (words_out && nPath) memcpy(words_out, d->h_words.p, sizeof(unsigned) * nPath);
    if (getenv("DSR_VITERBI_SEG_VERBOSE") && d->d_poolNext.p && d->lastPoolCap > 0) {
      unsigned long long used = 0; DSR_HIP(hipMemcpy(&used, d->d_poolNext.p, sizeof(used), hipMemcpyDeviceToHost));
      fprintf(stderr, "[dsr viterbi] pool of back-pointer records: %.1f M of %.1f M taken\n", used / 1e6, d->lastPoolCap / 1e6);
    }
    if (prof) {
      std::vector<long long> hp((size_t) slots * 32); DSR_HIP(hipMemcpy(hp.data(), prof, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
      double acc[32] = {0}; for (int s2 = 0; s2 < slots; s2++) for (int i = 0; i < 32; i++) acc[i] += (double) hp[(size_t) s2 * 32 + i];
      fprintf(stderr, "[dsr viterbi prof] mean us per slot:");
      for (int i = 0; i < 32; i++) if (i != 15) fprintf(stderr, " p%d=%.0f", i, acc[i] / slots / 100.0);
      fprintf(stderr, "\n");
      double tmin = 1e30, tmax = 0.0, tsum = 0.0;                   // busy time per slot: how even the slots' shares of the batch were
      for (int s2 = 0; s2 < slots; s2++) { double t = 0.0; for (int i = 0; i < 32; i++) if (i != 15) t += (double) hp[(size_t) s2 * 32 + i]; t /= 100.0; tsum += t; if (t < tmin) tmin = t; if (t > tmax) tmax = t; }
      fprintf(stderr, "[dsr viterbi prof] busy us per slot: mean %.0f min %.0f max %.0f\n", tsum / slots, tmin, tmax);
      if (slots >= 64 && slots % 8 == 0) {                       // by XCD (workgroup i runs on XCD i mod 8): how even the eight queues of the time-sliced decode come out
        double bx[8] = {0}; for (int s2 = 0; s2 < slots; s2++) { double t = 0.0; for (int i = 0; i < 32; i++) if (i != 15) t += (double) hp[(size_t) s2 * 32 + i]; bx[s2 & 7] += t / 100.0; }
        fprintf(stderr, "[dsr viterbi prof] busy us per slot, by XCD:"); for (int q = 0; q < 8; q++) fprintf(stderr, " %.0f", bx[q] / (slots / 8)); fprintf(stderr, "\n");
      }
    }
    if (d->dumpOn) {
      long cnt[2]; DSR_HIP(hipMemcpy(cnt, d->d_dumpCount.p, sizeof(cnt), hipMemcpyDeviceToHost));
      const long N = cnt[0] < d->dumpCap ? cnt[0] : d->dumpCap; d->h_dumpFrames = cnt[1];
      std::vector<long> fo(cnt[1] + 1); if (cnt[1] > 0) DSR_HIP(hipMemcpy(fo.data(), d->d_dumpFrameOff.p, sizeof(long) * (cnt[1] + 1), hipMemcpyDeviceToHost));
      d->h_dumpFrameOff.assign(fo.begin(), fo.end());
      d->h_dumpNode.resize(N); d->h_dumpArc.resize(N); d->h_dumpAc.resize(N); d->h_dumpLm.resize(N);
      if (N > 0) {
        DSR_HIP(hipMemcpy(d->h_dumpNode.data(), d->d_dumpNode.p, sizeof(int) * N, hipMemcpyDeviceToHost));
        DSR_HIP(hipMemcpy(d->h_dumpArc.data(), d->d_dumpArc.p, sizeof(int) * N, hipMemcpyDeviceToHost));
        DSR_HIP(hipMemcpy(d->h_dumpAc.data(), d->d_dumpAc.p, sizeof(float) * N, hipMemcpyDeviceToHost));
        DSR_HIP(hipMemcpy(d->h_dumpLm.data(), d->d_dumpLm.p, sizeof(float) * N, hipMemcpyDeviceToHost));
      }
    }
  });
}

dsr_status dsr_decoder_decode_batch(dsr_decoder* d, const float* score, const int32_t* nframes, int U, int Tmax, int nDist,
                                    dsr_decode_result* res, int32_t* arcs_out, uint32_t* words_out, int maxPath, void* stream)
{
  if (!res) return guard([&] { throw Error(DSR_E_PARAMETER, "null argument"); });
  if (U <= 0) return DSR_OK;
  const dsr_status s1 = dsr_decoder_decode_launch(d, score, nframes, U, Tmax, nDist, maxPath, (arcs_out || words_out) ? 1 : 0, stream);
  if (s1 != DSR_OK) return s1;
  return dsr_decoder_decode_collect(d, res, arcs_out, words_out);
}

// _Decoder::lattice() (decoder.h:805-860) for utterance u of the last decode (cfg.latticeTokens > 0)
// the placement log of utterance u of the last lattice-mode decode, copied to the host
namespace {
struct LatHost {
  std::vector<long> frameOff; std::vector<dsr::LatPlace> place; std::vector<double> ttl; std::vector<dsr::LatBp> arena; std::vector<int> arenaLat; std::vector<dsr::LatFinalTok> fin;
  dsr::LatInput in;
};
void load_lat(dsr_decoder* d, int u, uint32_t eosX, LatHost& H)
{
  if (d->latU <= 0) throw Error(DSR_E_CONSISTENCY, "Must enable lattice generation during decoding.");                 // decoder.h:807-808
  if (d->pendingU > 0) throw Error(DSR_E_CONSISTENCY, "a decode is in flight: collect it first");
  if (u < 0 || u >= d->latU) throw Error(DSR_E_INDEX, "utterance %d of %d", u, d->latU);
  int info[4]; DSR_HIP(hipMemcpy(info, d->d_latInfo.p + 4 * (size_t) u, sizeof(info), hipMemcpyDeviceToHost));
  const int finN = info[0], haveNext = info[1], T = info[3]; const long arenaN = info[2];
  if (T <= 0) throw Error(DSR_E_CONSISTENCY, "utterance %d was not decoded to its end (status of its decode result)", u);
  H.frameOff.assign((size_t) T + 2, 0);
  DSR_HIP(hipMemcpy(H.frameOff.data(), d->d_latFrameOff.p + (size_t) u * (d->latTmax + 3), sizeof(long) * ((size_t) T + 2), hipMemcpyDeviceToHost));
  const long nP = H.frameOff[(size_t) T + 1];
  H.place.resize((size_t) (nP > 0 ? nP : 1)); H.ttl.resize((size_t) (nP > 0 ? nP : 1));
  H.arena.resize((size_t) (arenaN > 0 ? arenaN : 1)); H.arenaLat.resize((size_t) (arenaN > 0 ? arenaN : 1)); H.fin.resize((size_t) (finN > 0 ? finN : 1));
  const size_t cap = (size_t) d->cfg.latticeTokens;
  if (nP > 0) { DSR_HIP(hipMemcpy(H.place.data(), d->d_lat.p + (size_t) u * cap, sizeof(LatPlace) * (size_t) nP, hipMemcpyDeviceToHost));
                DSR_HIP(hipMemcpy(H.ttl.data(), d->d_latTtl.p + (size_t) u * cap, sizeof(double) * (size_t) nP, hipMemcpyDeviceToHost)); }
  if (arenaN > 0) { DSR_HIP(hipMemcpy(H.arena.data(), d->d_arena.p + (size_t) u * (size_t) d->latArenaCap, sizeof(LatBp) * (size_t) arenaN, hipMemcpyDeviceToHost));
                    DSR_HIP(hipMemcpy(H.arenaLat.data(), d->d_arenaLat.p + (size_t) u * (size_t) d->latArenaCap, sizeof(int) * (size_t) arenaN, hipMemcpyDeviceToHost)); }
  if (finN > 0) DSR_HIP(hipMemcpy(H.fin.data(), d->d_latFinal.p + (size_t) u * d->cfg.maxActive, sizeof(LatFinalTok) * (size_t) finN, hipMemcpyDeviceToHost));
  LatInput& in = H.in; in.graph = &d->graphCopy; in.csr = &d->csr; in.tab = &d->tab; in.lmScale = d->cfg.lmScale; in.lmPenalty = d->cfg.lmPenalty; in.silPenalty = d->cfg.silPenalty;
  in.silenceX = d->cfg.silenceX; in.eosX = eosX; in.T = T; in.place = H.place.data(); in.ttl = H.ttl.data(); in.frameOff = H.frameOff.data();
  in.arena = H.arena.data(); in.arenaLat = H.arenaLat.data(); in.arenaN = arenaN; in.fin = H.fin.data(); in.finN = finN; in.haveNext = haveNext;
}
}  // namespace

dsr_status dsr_decoder_lattice(dsr_decoder* d, int u, uint32_t eosX, dsr_lattice** out)
{
  return guard([&] {
    if (!d || !out) throw Error(DSR_E_PARAMETER, "null argument");
    LatHost H; load_lat(d, u, eosX, H);
    dsr_lattice* L = new dsr_lattice();
    try { build_lattice(H.in, *L); } catch (...) { delete L; throw; }
    *out = L;
  });
}

// _Decoder::writeGMM(conv, channel, spk, utt, cfrom, score, fileName, frameInterval) (decoder.h:1018-1102; decoder.i:177-178): the 1-best path as
// runs of equal input symbols -- "# utt cfrom score", then per run "conv channel start duration label score", first run first, the end-of-sentence
// label skipped.  The walk needs every token of the best path with its frame and scores: the decoder must have run with lattice bookkeeping
// (latticeTokens > 0, the reference's generateLattice) and symbols set (dsr_decoder_set_symbols: labels come from the input lexicon, :1066).
// fileName "" or NULL: stdout; files are appended to (:1023).  (spk is accepted and unused, as in the reference.)
dsr_status dsr_decoder_write_gmm(dsr_decoder* d, int u, const char* conv, const char* channel, const char* spk, const char* utt, double cfrom, double score,
                                 const char* fileName, double frameInterval)
{
  return guard([&] {
    (void) spk;
    if (!d || !conv || !channel || !utt) throw Error(DSR_E_PARAMETER, "null argument");
    if (!d->lexIn) throw Error(DSR_E_KEY, "the transducer set on this decoder has no input lexicon");
    LatHost H; load_lat(d, u, 0u, H);
    std::vector<GmmRow> rows;
    if (!best_path_gmm(H.in, rows)) throw Error(DSR_E_CONSISTENCY, "no best token");       // (the reference dereferences a null token here)
    FILE* fp = (!fileName || !*fileName) ? stdout : fopen(fileName, "a");
    if (!fp) throw Error(DSR_E_IO, "could not open %s", fileName);
    fprintf(fp, "# %s %10.4f %10.4f\n", utt, cfrom, score);
    for (int i = (int) rows.size() - 1; i >= 0; i--) {
      if (rows[i].inX >= d->lexIn->syms.size()) { if (fp != stdout) fclose(fp); throw Error(DSR_E_INDEX, "input symbol %u is not in the lexicon", rows[i].inX); }
      const std::string& lab = d->lexIn->symbol(rows[i].inX); const char* label = lab.c_str();
      if (lab == d->eosSymbol) continue;
      const double beg = cfrom + rows[i].startX * frameInterval, len = (rows[i].endX - rows[i].startX + 1) * frameInterval;
      fprintf(fp, "%s %s %7.2f %7.2f %-20s %7.2f\n", conv, channel, beg, len, label, rows[i].score);
    }
    if (fp != stdout) fclose(fp); else fflush(stdout);
  });
}
void dsr_lattice_destroy(dsr_lattice* L) { delete L; }
int dsr_lattice_num_nodes(const dsr_lattice* L) { return L ? (int) L->nodeFinal.size() : 0; }
int dsr_lattice_num_edges(const dsr_lattice* L) { return L ? (int) L->from.size() : 0; }
int dsr_lattice_final_states_n(const dsr_lattice* L) { return L ? L->finalStatesN : 0; }
dsr_status dsr_lattice_get(const dsr_lattice* L, int32_t* nodeFinal, int32_t* from, int32_t* to, uint32_t* in, uint32_t* out, int32_t* start, int32_t* end, double* ac, double* lm)
{
  return guard([&] {
    if (!L) throw Error(DSR_E_PARAMETER, "null argument");
    const size_t nE = L->from.size();
    if (nodeFinal) memcpy(nodeFinal, L->nodeFinal.data(), 4 * L->nodeFinal.size());
    if (from && nE) memcpy(from, L->from.data(), 4 * nE); if (to && nE) memcpy(to, L->to.data(), 4 * nE);
    if (in && nE) memcpy(in, L->in.data(), 4 * nE); if (out && nE) memcpy(out, L->out.data(), 4 * nE);
    if (start && nE) memcpy(start, L->start.data(), 4 * nE); if (end && nE) memcpy(end, L->end.data(), 4 * nE);
    if (ac && nE) memcpy(ac, L->ac.data(), 8 * nE); if (lm && nE) memcpy(lm, L->lm.data(), 8 * nE);
  });
}
dsr_status dsr_lattice_write(dsr_lattice* L, const char* fileName, int writeData)
{ return guard([&] { if (!L || !fileName) throw Error(DSR_E_PARAMETER, "null argument"); L->write(fileName, writeData != 0); }); }
size_t dsr_lattice_pack_size(const dsr_lattice* L) { return L ? 16 + 4 * L->nodeFinal.size() + 40 * L->from.size() : 0; }
dsr_status dsr_lattice_pack(const dsr_lattice* L, void* buf, size_t bufBytes)
{
  return guard([&] {
    if (!L || !buf) throw Error(DSR_E_PARAMETER, "null argument");
    const std::vector<unsigned char> b = L->pack();
    if (b.size() > bufBytes) throw Error(DSR_E_DIMENSION, "buffer holds %zu bytes, the lattice needs %zu", bufBytes, b.size());
    memcpy(buf, b.data(), b.size());
  });
}
dsr_status dsr_lattice_unpack(const void* buf, size_t bytes, dsr_lattice** out)
{
  return guard([&] {
    if (!buf || !out) throw Error(DSR_E_PARAMETER, "null argument");
    dsr_lattice* L = new dsr_lattice();
    try { static_cast<LatticeData&>(*L) = LatticeData::unpack((const unsigned char*) buf, bytes); } catch (...) { delete L; throw; }
    *out = L;
  });
}

dsr_status dsr_decoder_get_dump(dsr_decoder* d, int64_t* nFrames, const int64_t** frameOff, const int32_t** node,
                                const float** ac, const float** lm, const int32_t** arc)
{
  return guard([&] {
    if (!d) throw Error(DSR_E_PARAMETER, "null argument");
    if (nFrames) *nFrames = d->h_dumpFrames;
    if (frameOff) *frameOff = d->h_dumpFrameOff.data();
    if (node) *node = d->h_dumpNode.data(); if (arc) *arc = d->h_dumpArc.data();
    if (ac) *ac = d->h_dumpAc.data(); if (lm) *lm = d->h_dumpLm.data();
  });
}

}  // extern "C"
