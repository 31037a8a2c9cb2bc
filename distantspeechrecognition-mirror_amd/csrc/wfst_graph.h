// csrc/wfst_graph.h -- host-side static decoding graph with the reference's container semantics
// plus the flattened tables the device decoder walks.
//
// Container semantics follow WFSTFlyWeight (asr/decoder/wfstFlyWeight.cc):
//   - the source state of the first arc becomes the initial node, a node object of its own that
//     find() returns before consulting the maps (:94-118); _addFinal never consults it (:72-92)
//   - arcs are prepended to their source node (:552-556): iteration order = reverse file order
//   - the text reader drops epsilon:epsilon self loops (:353); the binary reader does not
// Device tables (new design): per node the list of placements a token in that node produces in
// one frame, in the exact order _Decoder::_expandNode (asr/decoder/decoder.h:956-989) makes them --
// depth-first through epsilon arcs, which the reference expands recursively without recombination.
#pragma once
#include "common.h"
#include <functional>
#include <string>
#include <unordered_map>

namespace dsr {

struct XRec { int32_t dst; int32_t dist; float cost; uint32_t meta; };      // meta: [15:0] path length, bit16: output != 0
struct ERec { int32_t dst; int32_t lastSrc; int32_t pathOff; int32_t pathLen; };

struct WfstGraph {
  struct Node { uint32_t state; int final_; float cost; int firstArc; bool inNodes, inFinal; };
  struct Arc { int src, dst; uint32_t in, out; float cost; int next; };
  std::vector<Node> nodes; std::vector<Arc> arcs;
  std::unordered_map<uint32_t, int> nodeOf;   // state -> node id in _nodes / _final (the initial node is not in it: wfstFlyWeight.cc:94-118)
  int initial = -1;
  // WFSTFlyWeightSortedOutput (wfstFlyWeight.h:403-424): a node keeps its arcs ordered by (output, input); a new arc goes in front of the first arc that
  // is not smaller (Node::_addEdgeForce, wfstFlyWeight.cc:754-776).  The container DecoderWordTrace searches (decoder.h:1146-1149).
  bool sortedOutput = false;
  // text files may name states and symbols instead of numbering them: a field that does not start with a number is looked up in the state (0),
  // input (1) or output (2) lexicon (wfstFlyWeight.cc:311-347); unset: such a field is an error
  std::function<uint32_t(int, const char*)> symbolOf;
  // the other direction, for write(useSymbols = true): which = 0 state, 1 input, 2 output lexicon; stateLexSize: statelex null or empty -> 0
  std::function<std::string(int, uint32_t)> nameOf; std::function<size_t()> stateLexSize;

  int  findNode(uint32_t state, bool create);
  void addFinal(uint32_t state, float cost);
  void addArc(uint32_t s1, uint32_t s2, uint32_t in, uint32_t out, float cost, bool dropEpsSelf);
  void read(const char* file, bool binary);
  void readEx(const char* file, bool binary, bool noSelfLoops);
  void write(const char* file, bool binary, bool useSymbols = false) const;
  void reverse(const WfstGraph& src);                      // WFSTFlyWeight::reverse (wfstFlyWeight.cc:141-213)
  void reverseRead(const char* file);                      // WFSTFlyWeight::reverseRead (:215-297)
  void clear();                                            // _clear()
  int  lookup(uint32_t state) const;                       // node id in _nodes / _final or -1 (no initial-node shortcut)
  void addEdgeForce(int from, int to, uint32_t in, uint32_t out, float cost);   // Node::_addEdgeForce (:552-556): prepend

  // CSR in iteration order (node 0.. in creation order; the initial node is id `initial`)
  struct Csr { std::vector<int> off, dst; std::vector<uint32_t> in, out; std::vector<float> cost; std::vector<int> csrOf; };
  Csr csr() const;

  // flattened expansion tables
  struct Tables {
    std::vector<int> xoff; std::vector<XRec> xrec; std::vector<int> xarc, xpathOff;
    std::vector<int> eoff; std::vector<ERec> erec;
    std::vector<int> path;             // epsilon arc ids (CSR numbering)
  };
  Tables tables(const Csr& c, size_t maxRecords) const;
};

}  // namespace dsr
