// csrc/wordtrace.h -- what the decoder object (k_viterbi.hip) hands to the DecoderWordTrace search kernel (k_wordtrace.hip)
#pragma once
#include "common.h"
#include "wfst_graph.h"

namespace dsr {

struct WTok { float ac, lm; int node, wt, arc; };            // a token: scores as the floats a Token holds, its state, its word trace (-1: none), its edge (CSR arc)
struct WCand { float ac, lm; int dst, wt, arc; };            // a placement

struct WtArgs {
  // the transducer's expansion tables (wfst_graph.h), CSR numbering
  int nNodes, initial;
  const int* xoff; const XRec* xrec; const int* xarc; const int* xpathOff; const int* eoff; const ERec* erec; const int* path;
  const float* arcCost; const uint32_t* arcOut; const uint32_t* arcIn; const int* nodeFinal; const float* nodeCost;
  // the decoder's settings (decoder.i:201-260)
  double beam, lmScale, lmPenalty, silPenalty; uint32_t silenceX; int insertSilence;
  int maxTok, maxCand; long maxTraces;
  // scratch: per slot token lists (2 x maxTok), placements, offsets, flags; per slot and state the first-arrival slot and the best (score, slot) key
  // (all ones between frames); per utterance the word traces {word, previous trace, frame, -}
  WTok* tok; WCand* cand; int* tokOff; int* rank; unsigned long long* bestKey; unsigned* firstSlot; int4* traces; int* queue;
  // input / output of dsr_decoder_decode_launch
  const float* scores; const int* nframes; int U, Tmax, nDist; dsr_decode_result* res; int* arcsOut; unsigned* wordsOut; int maxPath;
};

void wordtrace_launch(const WtArgs& A, int slots, hipStream_t st);

}  // namespace dsr
