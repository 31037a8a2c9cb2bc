// csrc/lattice.h -- lattice generation from the decoder's placement log (host side).
// Replaces _Decoder::lattice / _majorTrace / _minorTrace / _findLNode (asr/decoder/decoder.h:805-953) on top of the 'worse' chains that
// _placeOnList builds (decoder.h:531-541), and Lattice::write (asr/lattice/lattice.cc:715-757).
#pragma once
#include "wfst_graph.h"
#include <vector>

namespace dsr {

struct LatBp { uint32_t prev, rec; };                      // one back-pointer record of the decoder kernel
struct LatPlace { uint32_t ac, lm, rec, prevBp; };         // one placement as the kernel logs it (float bits, record, parent back pointer)
struct LatFinalTok { int32_t node, bp, ac, lm; };

struct LatInput {
  const WfstGraph* graph; const WfstGraph::Csr* csr; const WfstGraph::Tables* tab;
  double lmScale, lmPenalty, silPenalty; uint32_t silenceX, eosX;
  int T;                                                   // frames of the utterance; bucket f < T: frame f, bucket T: the end expansion
  const LatPlace* place; const double* ttl; const long* frameOff;   // frameOff[T + 2]
  const LatBp* arena; const int* arenaLat; long arenaN;
  const LatFinalTok* fin; int finN; int haveNext;          // _next after _expandToEnd (haveNext) or _current
};

struct LatticeData {
  std::vector<int> nodeFinal;                              // per lattice node index (0 = initial): 1 final, 0 not
  std::vector<int> from, to, start, end; std::vector<uint32_t> in, out; std::vector<double> ac, lm;   // edges in creation order
  int finalStatesN = 0;
  void write(const char* file, bool writeData) const;      // Lattice::write(file, useSymbols = false, writeData)
  std::vector<unsigned char> pack() const;                 // flat little-endian image for the gather across ranks
  static LatticeData unpack(const unsigned char* p, size_t n);
};

void build_lattice(const LatInput& in, LatticeData& out);

struct GmmRow { uint32_t inX; int startX, endX; double score; };
bool best_path_gmm(const LatInput& in, std::vector<GmmRow>& rows);      // false: no best token

}  // namespace dsr
