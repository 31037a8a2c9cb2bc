// csrc/lattice.h -- lattice generation from the decoder's placement log (host side).
// Replaces _Decoder::lattice / _majorTrace / _minorTrace / _findLNode (asr/decoder/decoder.h:805-953) on top of the 'worse' chains that
// _placeOnList builds (decoder.h:531-541), and Lattice::write (asr/lattice/lattice.cc:715-757).
#pragma once
#include "wfst_graph.h"
#include <utility>
#include <vector>

struct dsr_lexicon;

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
  std::vector<int> nodeFinal;                              // per lattice node (0 = initial, then in creation order): 1 final, 0 not
  std::vector<int> from, to, start, end; std::vector<uint32_t> in, out; std::vector<double> ac, lm;   // edges in creation order
  int finalStatesN = 0;
  void write(const char* file, bool writeData);            // Lattice::write(file, useSymbols = false, writeData)
  std::vector<unsigned char> pack() const;                 // flat little-endian image for the gather across ranks (the structure as built)
  static LatticeData unpack(const unsigned char* p, size_t n);

  // ---- the asr/lattice operations (asr/lattice/lattice.cc): lattice_ops.cpp.  The object keeps what the reference's object keeps between
  // calls: the rescoring tokens on the nodes, forward/backward probabilities, link posteriors, the cached topological order, node colours,
  // the three places a node can live in (initial / _nodes[index] / _final[key]) and the index a node prints under (renumbered by prune/purge).
  struct Tok { float ac, lm; int edge, prev; };            // _Token: float scores (lattice.h:73-74), edge, prev
  std::vector<int> index;                                  // printed index per node
  std::vector<float> ecost, ncost;                         // Weight of an edge / of a final node (0 for decoder-built lattices)
  std::vector<double> gamma, fwd, bwd;                     // per edge; per node
  std::vector<std::vector<int> > adj;                      // a node's edge list, head first (the last edge added), pruned links taken out
  std::vector<int> slots;                                  // _nodes: node or -1
  std::vector<std::pair<unsigned, int> > finals;           // _final: key -> node, key order
  std::vector<int> color, success, ntok, sorted; std::vector<Tok> toks;
  double acScale = 1.0, lmScale = 1.0, lmPenalty = 0.0, silPenalty = 0.0, latticeForwardProb = 0.0; unsigned silenceX = 0;   // lattice.cc:64-69
  bool opsReady = false;
  void ensure_ops();
  void topo_sort();                                        // _topoSort/_visitNode (lattice.cc:858-887): cached until _clearSorted
  int best_token() const;                                  // _bestToken (:261-279); -1: none
  float rescore(double lmScale, double lmPenalty, double silPenalty, unsigned silenceX);                          // :122-134
  std::vector<unsigned> best_hypo(bool useInputSymbols) const;                                                    // :281-306, first symbol first
  double gamma_probs(double acScale, double lmScale, double lmPenalty, double silPenalty, unsigned silenceX);     // :309-379
  void prune(double threshold);                            // :648-693
  void prune_edges(unsigned edgesN);                       // :695-713
  void purge();                                            // :830-841, 776-828
  static LatticeData read(const char* file, bool noSelfLoops, bool readData, struct ::dsr_lexicon* inlex, struct ::dsr_lexicon* outlex);   // fsm.h:3787-3873
};

void build_lattice(const LatInput& in, LatticeData& out);

struct GmmRow { uint32_t inX; int startX, endX; double score; };
bool best_path_gmm(const LatInput& in, std::vector<GmmRow>& rows);      // false: no best token

}  // namespace dsr

struct dsr_lattice : dsr::LatticeData {};
