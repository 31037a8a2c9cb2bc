// csrc/k_wordtrace.hip -- the search of DecoderWordTrace (asr/decoder/decoder.h:1146-1304, asr/decoder/decoder.cc:126-470) with generateLattice = false,
// batched over utterances.
//
// What this search does differently from _Decoder<> (k_viterbi.hip), and why it is a kernel of its own:
//   * a token's scores are rounded to float BEFORE it is compared: _advanceTokens builds the Token (decoder.cc:357-392), _placeOnList then compares
//     tok->score() -- the float sum of the two floats -- with the incumbent's (:215,262).  Recombination is therefore "smallest float score, the
//     earliest arrival among equals": an order-free minimum over (score, arrival slot), no replay of an order-dependent fold;
//   * the edge's language-model increment is formed first (lmScale x cost + lmScale x lmPenalty), then added to the token's, then the silence penalty
//     (:398-412, 371-376) -- another rounding order than _expandNode's (decoder.h:956-989);
//   * tokens carry no back pointer but a WORD TRACE: crossing an arc with an output symbol (or, with insertSilence, entering silence) makes a new
//     WordTrace {wordX, wordSequenceX, frame, the token at the boundary} (decoder.cc:421-428); the hypothesis is the chain of traces.  The traces live in
//     a per-utterance arena {word, previous trace, frame}; wordSequenceX is only read by the generateLattice merge (:239-249) and is not kept;
//   * the transducer is a WFSTFlyWeightSortedOutput (arcs of a node ordered by (output, input): wfst_graph.cpp addEdgeForce);
//   * the end expansion rounds the final-state cost to float first ("float lmScoreEdge", decoder.cc:192,454).
// One 256-thread workgroup per utterance, persistent over frames, placements staged in memory, first-arrival slot and best (score, slot) key per state
// by two atomic minima.  Feature parity, not speed: the reference ships this class without a driver.  generateLattice = true is refused by the caller
// (undefined in the reference on any graph whose first arcs carry no output symbol: _placeOnList dereferences a null word trace, :239).
// Compiled with -ffp-contract=off.
#include "common.h"
#include "wfst_graph.h"
#include "wordtrace.h"
#include <cmath>

namespace dsr {

static constexpr int kWT = 256;

// the two per-state tables are written by atomics (executed at L2) and by the restoring stores: they are read and restored past the L1 too
__device__ __forceinline__ unsigned wt_ld32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long wt_ld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wt_reset(unsigned* f, unsigned long long* b) { __hip_atomic_store(f, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(b, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned wt_f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// block-wide exclusive prefix sum of one int per thread (kWT threads); total in *tot
__device__ __forceinline__ int wt_block_excl(int v, int* s_w, int& tot)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
  __syncthreads();
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int base = 0, t = 0;
  for (int w = 0; w < kWT / 64; w++) { const int q = s_w[w]; if (w < wave) base += q; t += q; }
  tot = t;
  return base + incl - v;
}

__global__ __launch_bounds__(kWT) void k_wordtrace(const WtArgs A)
{
  __shared__ int s_w[kWT / 64]; __shared__ double s_min[kWT / 64]; __shared__ unsigned long long s_key[kWT / 64];
  __shared__ int s_u, s_trace, s_status;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slot = blockIdx.x;
  WTok* tok0 = A.tok + (size_t) slot * 2 * A.maxTok; WTok* tok1 = tok0 + A.maxTok;
  WCand* cand = A.cand + (size_t) slot * A.maxCand;
  int* tokOff = A.tokOff + (size_t) slot * (A.maxTok + 1);
  int* flagR = A.rank + (size_t) slot * A.maxCand;
  unsigned long long* bestKey = A.bestKey + (size_t) slot * A.nNodes; unsigned* firstSlot = A.firstSlot + (size_t) slot * A.nNodes;
  const double lsPen = __dmul_rn(A.lmScale, A.lmPenalty), lsSil = __dmul_rn(A.lmScale, A.silPenalty);

  for (;;) {
    __syncthreads();
    if (tid == 0) { s_u = atomicAdd(A.queue, 1); s_trace = 0; s_status = DSR_OK; }
    __syncthreads();
    const int u = s_u;
    if (u >= A.U) break;
    const int T = A.nframes[u] < A.Tmax ? A.nframes[u] : A.Tmax;
    const float* sc = A.scores + (size_t) u * A.Tmax * A.nDist;
    int4* traces = A.traces + (size_t) u * A.maxTraces;
    WTok* cur = tok0; WTok* nxt = tok1;
    int n = 0; long activeHypos = 0; double topScore = HUGE_VAL, thresh = HUGE_VAL;
    int status = (T <= 0) ? DSR_E_ITERATOR : DSR_OK;
    int numNew = 0;

    // _advanceTokens for one token (the worse chain has one element without generateLattice) + the word boundary (decoder.cc:357-392, 414-428):
    // ac / lm in and out as the FLOATS a Token holds; inSil: the token's edge input == silenceX; has: there is a token (not the start of the utterance)
    auto advance = [&](const int arc, const double acEdge, double lmEdge, const bool has, float& ac, float& lm, bool& inSil, int& wt, const int frameX, const bool wordRule) {
      const unsigned ain = A.arcIn[arc], aout = A.arcOut[arc];
      double a2, l2;
      if (!has) { if (ain == A.silenceX) lmEdge = __dadd_rn(lmEdge, lsSil); a2 = acEdge; l2 = lmEdge; }
      else {
        a2 = __dadd_rn(acEdge, (double) ac); l2 = __dadd_rn(lmEdge, (double) lm);
        if (ain == A.silenceX && !inSil) l2 = __dadd_rn(l2, lsSil);
      }
      bool newWord = aout != 0u;
      if (wordRule && A.insertSilence && ain == A.silenceX && (!has || !inSil)) newWord = true;
      ac = (float) a2; lm = (float) l2; inSil = (ain == A.silenceX);
      if (newWord) {                                                         // new WordTrace(wordX, seq, frameX, wordToken): the boundary token's own trace is `wt`
        const int id = atomicAdd(&s_trace, 1);
        if (id < A.maxTraces) traces[id] = make_int4((int) aout, wt, frameX, 0); else s_status = DSR_E_ALLOCATION;
        wt = id;
      }
    };

    for (int fr = 0; fr <= T && status == DSR_OK; fr++) {
      const bool endPhase = (fr == T);
      const int frameX = endPhase ? T - 1 : fr;                              // decode(): _frameX-- before _expandToEnd (decoder.h:716-717)
      const bool first = (fr == 0);
      // ---- placements per token: _expandNode's emitting arcs with their epsilon paths, or (end) the final-state placements
      const int nTok = first ? 1 : n;
      const int per = (nTok + kWT - 1) / kWT;                                // a contiguous chunk of the list per thread: the scan keeps list order
      int loc = 0;
      for (int q = 0; q < per; q++) {
        const int i = tid * per + q; int cnt = 0;
        if (i < nTok) {
          const int node = first ? A.initial : cur[i].node;
          bool live = true;
          if (!first && !endPhase) live = !((double) __fadd_rn(cur[i].ac, cur[i].lm) > thresh);      // beam (decoder.cc:161)
          if (live) cnt = endPhase ? (A.nodeFinal[node] ? 1 : 0) + (A.eoff[node + 1] - A.eoff[node]) : A.xoff[node + 1] - A.xoff[node];
          flagR[i] = cnt;                                                      // (flagR doubles as the per-token count until the scan below)
        }
        loc += cnt;
      }
      int C; int base = wt_block_excl(loc, s_w, C);
      for (int q = 0; q < per; q++) { const int i = tid * per + q; if (i < nTok) { const int cnt = flagR[i]; tokOff[i] = base; base += cnt; } }
      if (tid == 0) tokOff[nTok] = C;
      __syncthreads();
      if (C > A.maxCand) { status = DSR_E_ALLOCATION; break; }
      // ---- one thread per placement
      double locMin = HUGE_VAL;
      for (int c = tid; c < C; c += kWT) {
        int lo = 0, hi = nTok;                                               // last token whose offset is <= c (tokens without placements share their successor's offset)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tokOff[mid] <= c) lo = mid; else hi = mid; }
        const int i = lo, j = c - tokOff[i];
        const bool has = !first;
        float ac = has ? cur[i].ac : 0.0f, lm = has ? cur[i].lm : 0.0f; int wt = has ? cur[i].wt : -1;
        bool inSil = has ? (A.arcIn[cur[i].arc] == A.silenceX) : false;
        const int node = first ? A.initial : cur[i].node;
        int dst, lastArc; bool h2 = has;
        if (!endPhase) {
          const int rec = A.xoff[node] + j; const XRec x = A.xrec[rec];
          const int plen = (int) (x.meta & 0xFFFFu); const int* pp = A.path + A.xpathOff[rec];
          for (int h = 0; h < plen; h++) {                                   // epsilon arcs: _expandNode recursion (decoder.cc:430-431)
            const int a = pp[h];
            double lmEdge = __dmul_rn(A.lmScale, (double) A.arcCost[a]);
            if (A.arcOut[a] != 0u) lmEdge = __dadd_rn(lmEdge, lsPen);
            advance(a, 0.0, lmEdge, h2, ac, lm, inSil, wt, frameX, true); h2 = true;
          }
          const int a = A.xarc[rec];
          double lmEdge = __dmul_rn(A.lmScale, (double) x.cost);
          if (A.arcOut[a] != 0u) lmEdge = __dadd_rn(lmEdge, lsPen);
          advance(a, (double) sc[(size_t) fr * A.nDist + x.dist], lmEdge, h2, ac, lm, inSil, wt, frameX, true);
          dst = x.dst; lastArc = a;
        } else {
          const int hasSelf = A.nodeFinal[node] ? 1 : 0;
          if (hasSelf && j == 0) {                                           // _expandToEnd (decoder.cc:190-196): the token's own edge again, float cost
            const float lmF = (float) __dmul_rn(A.lmScale, (double) A.nodeCost[node]);
            ac = (float) __dadd_rn(0.0, (double) ac); lm = (float) __dadd_rn((double) lmF, (double) lm);
            dst = node; lastArc = cur[i].arc;
          } else {
            const ERec e = A.erec[A.eoff[node] + (j - hasSelf)]; const int* pp = A.path + e.pathOff;
            for (int h = 0; h < e.pathLen; h++) {                            // _expandNodeToEnd (decoder.cc:436-467): no insertSilence rule here
              const int a = pp[h];
              double lmEdge = __dmul_rn(A.lmScale, (double) A.arcCost[a]);
              if (A.arcOut[a] != 0u) lmEdge = __dadd_rn(lmEdge, lsPen);
              advance(a, 0.0, lmEdge, true, ac, lm, inSil, wt, frameX, false);
              lastArc = a;
            }
            const float lmF = (float) __dmul_rn(A.lmScale, (double) A.nodeCost[e.dst]);
            const int a = pp[e.pathLen - 1];                                 // endToken = _advanceTokens(edge, 0, lmScoreFinal, wordToken): same edge, no new trace
            { const unsigned ain = A.arcIn[a]; double l2 = __dadd_rn((double) lmF, (double) lm); if (ain == A.silenceX && !inSil) l2 = __dadd_rn(l2, lsSil);
              ac = (float) __dadd_rn(0.0, (double) ac); lm = (float) l2; }
            dst = e.dst; lastArc = a;
          }
        }
        const float s = __fadd_rn(ac, lm);
        WCand cd; cd.ac = ac; cd.lm = lm; cd.dst = dst; cd.wt = wt; cd.arc = lastArc; cand[c] = cd;
        if (!endPhase && (double) s < locMin) locMin = (double) s;           // _topScore: placements over arcs with an input symbol (decoder.cc:217)
        atomicMin(&firstSlot[dst], (unsigned) c);
        atomicMin(&bestKey[dst], ((unsigned long long) wt_f2ord(s) << 32) | (unsigned) c);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(locMin, d, 64); locMin = (o < locMin) ? o : locMin; }
      if (lane == 0) s_min[wave] = locMin;
      __threadfence_block();
      __syncthreads();
      if (s_status != DSR_OK) { status = s_status; break; }
      if (!endPhase) { topScore = HUGE_VAL; for (int w = 0; w < kWT / 64; w++) if (s_min[w] < topScore) topScore = s_min[w]; }
      // ---- new list: states in reverse first-arrival order (_TokenList::insert prepends, replace keeps the place), each with its best arrival
      const int perC = (C + kWT - 1) / kWT;
      int nf = 0;
      for (int q = 0; q < perC; q++) { const int c = tid * perC + q; if (c < C) { const int f = (wt_ld32(&firstSlot[cand[c].dst]) == (unsigned) c) ? 1 : 0; flagR[c] = f; nf += f; } }
      int rbase = wt_block_excl(nf, s_w, numNew);
      if (numNew > A.maxTok) { status = DSR_E_ALLOCATION; break; }
      for (int q = 0; q < perC; q++) {
        const int c = tid * perC + q;
        if (c < C && flagR[c]) {
          const int dst = cand[c].dst; const unsigned w = (unsigned) (wt_ld64(&bestKey[dst]) & 0xFFFFFFFFull);
          const WCand cw = cand[w];
          WTok t; t.ac = cw.ac; t.lm = cw.lm; t.node = dst; t.wt = cw.wt; t.arc = cw.arc;
          nxt[numNew - 1 - rbase] = t; rbase++;
        }
      }
      __syncthreads();
      for (int c = tid; c < C; c += kWT) { const int dst = cand[c].dst; wt_reset(&firstSlot[dst], &bestKey[dst]); }     // the tables as they were
      __syncthreads();
      if (!endPhase) {
        if (numNew == 0) { status = DSR_E_CONSISTENCY; break; }
        { WTok* t = cur; cur = nxt; nxt = t; } n = numNew; activeHypos += numNew;
        thresh = __dadd_rn(topScore, A.beam);
      }
    }
    // ---- best token (decoder.h:639-685): _next after _expandToEnd, else _current; list order, strict '<' on the float score
    if (status == DSR_OK) {
      const WTok* lst = numNew > 0 ? nxt : cur; const int cntL = numNew > 0 ? numNew : n;
      unsigned long long key = ~0ull;
      for (int i = tid; i < cntL; i += kWT) { const float s = __fadd_rn(lst[i].ac, lst[i].lm); if (s == s) { const unsigned long long k = ((unsigned long long) wt_f2ord(s) << 32) | (unsigned) i; if (k < key) key = k; } }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(key, d, 64); key = (o < key) ? o : key; }
      if (lane == 0) s_key[wave] = key;
      __syncthreads();
      if (tid == 0) {
        unsigned long long k = ~0ull; for (int w = 0; w < kWT / 64; w++) if (s_key[w] < k) k = s_key[w];
        dsr_decode_result r; memset(&r, 0, sizeof(r));
        r.frames = T - 1; r.reachedFinal = numNew > 0 ? 1 : 0; r.finalStatesN = numNew; r.activeHypos = activeHypos; r.status = DSR_OK;
        if (k == ~0ull) r.status = DSR_E_CONSISTENCY;
        else {
          const WTok b = lst[(unsigned) (k & 0xFFFFFFFFu)];
          r.ac = b.ac; r.lm = b.lm; r.score = __dadd_rn((double) b.ac, (double) b.lm);
          // bestHypo (decoder.h:748-773): these tokens have no prev(): the walk is the best token's own edge
          r.nArcs = 1; if (A.arcsOut && A.maxPath > 0) A.arcsOut[(size_t) u * A.maxPath] = b.arc;
          // the hypothesis proper: the words along the word traces, first word first
          int nW = 0; for (int w = b.wt; w >= 0 && nW <= A.maxTraces; w = traces[w].y) nW++;
          r.nWords = nW;
          if (A.wordsOut) { int pos = nW; for (int w = b.wt; w >= 0 && pos > 0; w = traces[w].y) { pos--; if (pos < A.maxPath) A.wordsOut[(size_t) u * A.maxPath + pos] = (unsigned) traces[w].x; } }
          if (nW > A.maxPath && A.wordsOut) r.status = DSR_E_DIMENSION;
        }
        A.res[u] = r;
      }
    } else if (tid == 0) {
      dsr_decode_result r; memset(&r, 0, sizeof(r)); r.status = status; r.frames = T - 1; A.res[u] = r;
      // (an aborted frame leaves table entries behind: wiped below)
    }
    __syncthreads();
    if (status != DSR_OK) { for (int i = tid; i < A.nNodes; i += kWT) wt_reset(&firstSlot[i], &bestKey[i]); }
  }
}

void wordtrace_launch(const WtArgs& A, int slots, hipStream_t st)
{
  hipLaunchKernelGGL(k_wordtrace, dim3(slots), dim3(kWT), 0, st, A);
  DSR_HIP(hipGetLastError());
}

}  // namespace dsr
