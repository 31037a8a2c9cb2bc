// csrc/wfst_graph.cpp -- see wfst_graph.h.
#include "wfst_graph.h"
#include <algorithm>
#include <climits>
#include <cmath>

namespace dsr {

static const int kEndMarker = 2147483647;        // WFSTFlyWeight::EndMarker (wfstFlyWeight.cc:32)

static const uint32_t kMaximumIndex = 536870911u;  // WFSTFlyWeight::Node::_MaximumIndex (wfstFlyWeight.cc:522)

void WfstGraph::clear() { nodes.clear(); arcs.clear(); nodeOf.clear(); initial = -1; }

int WfstGraph::lookup(uint32_t state) const { auto it = nodeOf.find(state); return it == nodeOf.end() ? -1 : it->second; }

int WfstGraph::findNode(uint32_t state, bool create)
{
  if (initial >= 0 && nodes[initial].state == state) return initial;
  const int id = lookup(state);
  if (id >= 0) return id;
  if (!create) throw Error(DSR_E_KEY, "No state %u exists.", state);
  Node nd{state & 0x1FFFFFFFu, 0, 0.0f, -1, true, false};
  nodes.push_back(nd); nodeOf[state] = (int) nodes.size() - 1;
  return (int) nodes.size() - 1;
}

void WfstGraph::addFinal(uint32_t state, float cost)
{
  int id = lookup(state);
  if (id >= 0 && nodes[id].inFinal) throw Error(DSR_E_CONSISTENCY, "Automaton already has final node %d.", (int) state);
  if (id < 0) { Node nd{state & 0x1FFFFFFFu, 0, 0.0f, -1, false, false}; nodes.push_back(nd); id = (int) nodes.size() - 1; nodeOf[state] = id; }
  nodes[id].cost = cost; nodes[id].final_ = 1; nodes[id].inFinal = true; nodes[id].inNodes = false;
}

void WfstGraph::addEdgeForce(int from, int to, uint32_t in, uint32_t out, float cost)
{
  if (!sortedOutput) { Arc a{from, to, in, out, cost, nodes[from].firstArc}; arcs.push_back(a); nodes[from].firstArc = (int) arcs.size() - 1; return; }
  int ptr = nodes[from].firstArc, old = ptr;                        // WFSTFlyWeightSortedOutput::Node::_addEdgeForce (wfstFlyWeight.cc:754-776)
  while (ptr >= 0 && arcs[ptr].out < out) { old = ptr; ptr = arcs[ptr].next; }
  while (ptr >= 0 && arcs[ptr].out == out && arcs[ptr].in < in) { old = ptr; ptr = arcs[ptr].next; }
  Arc a{from, to, in, out, cost, -1}; arcs.push_back(a); const int id = (int) arcs.size() - 1;
  if (ptr == old) { arcs[id].next = nodes[from].firstArc; nodes[from].firstArc = id; }
  else { arcs[id].next = ptr; arcs[old].next = id; }
}

void WfstGraph::addArc(uint32_t s1, uint32_t s2, uint32_t in, uint32_t out, float cost, bool dropEpsSelf)
{
  int from;
  if (initial < 0) { Node nd{s1 & 0x1FFFFFFFu, 0, 0.0f, -1, false, false}; nodes.push_back(nd); initial = from = (int) nodes.size() - 1; }
  else from = findNode(s1, true);
  const int to = findNode(s2, true);
  if (dropEpsSelf && s1 == s2 && in == 0 && out == 0) return;
  addEdgeForce(from, to, in, out, cost);
}

// WFSTFlyWeight::reverse (wfstFlyWeight.cc:141-213): a super-initial node (index _MaximumIndex - 3) with an epsilon arc to every final node of the
// source carrying that node's cost, the source's initial state as the one final node, every arc turned round; nodes are looked up BY INDEX in the
// order the reference meets them (the arcs of the initial node, the final nodes, then the arcs of the final and of the internal nodes, both maps
// in key order), every turned arc prepended to its new source.
void WfstGraph::reverse(const WfstGraph& src)
{
  if (&src == this) throw Error(DSR_E_PARAMETER, "reverse: source and destination are the same transducer");
  if (src.initial < 0) throw Error(DSR_E_CONSISTENCY, "reverse: the source transducer is empty");
  clear();
  { Node nd{(kMaximumIndex - 3u) & 0x1FFFFFFFu, 0, 0.0f, -1, false, false}; nodes.push_back(nd); initial = 0; }     // initial(_MaximumIndex - 3)
  const uint32_t s0 = src.nodes[src.initial].state;
  addFinal(s0, 0.0f);
  const int rfinal = findNode(s0, false);
  for (int a = src.nodes[src.initial].firstArc; a >= 0; a = src.arcs[a].next) {                                     // from the final (i.e. initial) node
    const Arc& e = src.arcs[a]; const int r2 = findNode(src.nodes[e.dst].state, true);
    addEdgeForce(r2, rfinal, e.in, e.out, e.cost);
  }
  std::vector<std::pair<uint32_t, int>> fin, mid;
  for (size_t i = 0; i < src.nodes.size(); i++) { if (src.nodes[i].inFinal) fin.push_back(std::make_pair(src.nodes[i].state, (int) i)); if (src.nodes[i].inNodes) mid.push_back(std::make_pair(src.nodes[i].state, (int) i)); }
  std::sort(fin.begin(), fin.end()); std::sort(mid.begin(), mid.end());                                             // std::map order
  for (size_t k = 0; k < fin.size(); k++) {                                                                         // from the super initial node
    const Node& nd = src.nodes[fin[k].second]; const int rn = findNode(nd.state, true);
    addEdgeForce(initial, rn, 0, 0, nd.cost);
  }
  for (int pass = 0; pass < 2; pass++) {                                                                            // from the final nodes, then from the internal ones
    const std::vector<std::pair<uint32_t, int>>& ord = pass == 0 ? fin : mid;
    for (size_t k = 0; k < ord.size(); k++) {
      const Node& n1 = src.nodes[ord[k].second]; const int r1 = findNode(n1.state, true);
      for (int a = n1.firstArc; a >= 0; a = src.arcs[a].next) {
        const Arc& e = src.arcs[a]; const int r2 = findNode(src.nodes[e.dst].state, true);
        addEdgeForce(r2, r1, e.in, e.out, e.cost);
      }
    }
  }
}

// WFSTFlyWeight::reverseRead (wfstFlyWeight.cc:215-297): the text file read with every arc turned round.  A final-state line becomes an epsilon arc
// from the super-initial node to that state, which must exist by then (find without create: jkey_error otherwise); the source of the FIRST arc line
// becomes the final node (cost 0); "s1 == s2 && input == 0" arcs are dropped -- after their states have been created, and whatever their output.
void WfstGraph::reverseRead(const char* file)
{
  if (!file || !*file) throw Error(DSR_E_IO, "File name is null.");
  clear();
  { Node nd{(kMaximumIndex - 3u) & 0x1FFFFFFFu, 0, 0.0f, -1, false, false}; nodes.push_back(nd); initial = 0; }
  bool initialFlag = false;
  FILE* fp = fopen(file, "r");
  if (!fp) throw Error(DSR_E_IO, "Could not open file %s", file);
  char* line = nullptr; size_t cap = 0;
  try {
    while (getline(&line, &cap, fp) > 0) {
      char* tok[6]; int i = 0;
      tok[0] = strtok(line, " \t\n"); if (!tok[0]) continue;
      while ((i < 5) && ((tok[++i] = strtok(nullptr, " \t\n")) != nullptr));
      auto field = [&](int which, const char* t) -> uint32_t {
        char* p = nullptr; const unsigned long v = strtoul(t, &p, 0);
        if (p != t) return (uint32_t) v;
        if (!symbolOf) throw Error(DSR_E_KEY, "field '%s' is not a number and the transducer has no %s lexicon", t, which == 0 ? "state" : which == 1 ? "input" : "output");
        return symbolOf(which, t);
      };
      const uint32_t s1 = field(0, tok[0]);
      if (i == 1) addEdgeForce(initial, findNode(s1, false), 0, 0, 0.0f);
      else if (i == 2) { float c = 0.f; sscanf(tok[1], "%f", &c); addEdgeForce(initial, findNode(s1, false), 0, 0, c); }
      else if (i == 4 || i == 5) {
        const uint32_t s2 = field(0, tok[1]);
        if (!initialFlag) { addFinal(s1, 0.0f); initialFlag = true; }
        const int from = findNode(s1, true), to = findNode(s2, true);
        const uint32_t in = field(1, tok[2]), out = field(2, tok[3]);
        if (s1 == s2 && in == 0) continue;
        float c = 0.f; if (i == 5) sscanf(tok[4], "%f", &c);
        addEdgeForce(to, from, in, out, c);
      } else throw Error(DSR_E_IO, "Transducer file %s is inconsistent.", file);
    }
  } catch (...) { free(line); fclose(fp); throw; }
  free(line); fclose(fp);
}

namespace {
struct BEFile {
  FILE* fp;
  bool i32(int& v) { unsigned char b[4]; if (fread(b, 1, 4, fp) != 4) return false; v = (int) ((unsigned) b[0] << 24 | (unsigned) b[1] << 16 | (unsigned) b[2] << 8 | (unsigned) b[3]); return true; }
  int  i32x() { int v; if (!i32(v)) throw Error(DSR_E_IO, "Transducer binary file is inconsistent."); return v; }
  float f32x() { int i = i32x(); float f; memcpy(&f, &i, 4); return f; }
  void w32(int v) { unsigned u = (unsigned) v; unsigned char b[4] = { (unsigned char)(u >> 24), (unsigned char)(u >> 16), (unsigned char)(u >> 8), (unsigned char) u }; fwrite(b, 1, 4, fp); }
  void wf(float f) { int i; memcpy(&i, &f, 4); w32(i); }
};
}

void WfstGraph::read(const char* file, bool binary) { readEx(file, binary, false); }

// noSelfLoops: the dynamic container's reader, WFSTransducer::read(fileName, noSelfLoops) (asr/fsm/fsm.cc:901-986): every
// self loop is skipped before any node is looked up (:945); otherwise identical to the fly-weight text reader.
void WfstGraph::readEx(const char* file, bool binary, bool noSelfLoops)
{
  if (!file || !*file) throw Error(DSR_E_IO, "File name is null.");
  clear();                                                          // _clear()
  FILE* fp = fopen(file, binary ? "rb" : "r");
  if (!fp) throw Error(DSR_E_IO, "Could not open file %s", file);
  try {
    if (binary) {                                                   // _readBinary :367-413
      BEFile r{fp}; int n;
      while (r.i32(n) && n != kEndMarker) {
        if (n == 3) { const int idx = r.i32x(); const float c = r.f32x(); if (r.i32x() != kEndMarker) throw Error(DSR_E_IO, "Transducer binary file is inconsistent."); addFinal((uint32_t) idx, c); }
        else if (n == 6) { const int s1 = r.i32x(), s2 = r.i32x(), in = r.i32x(), out = r.i32x(); const float c = r.f32x();
          if (r.i32x() != kEndMarker) throw Error(DSR_E_IO, "Transducer binary file is inconsistent."); addArc((uint32_t) s1, (uint32_t) s2, (uint32_t) in, (uint32_t) out, c, false); }
        else throw Error(DSR_E_IO, "Transducer binary file is inconsistent.");
      }
    } else {                                                        // _readText :299-365 (numeric symbols)
      char* line = nullptr; size_t cap = 0;
      while (getline(&line, &cap, fp) > 0) {
        char* tok[6]; int i = 0;
        tok[0] = strtok(line, " \t\n"); if (!tok[0]) continue;
        while ((i < 5) && ((tok[++i] = strtok(nullptr, " \t\n")) != nullptr));
        auto field = [&](int which, const char* t) -> uint32_t {      // strtoul first, the lexicon when no digits were consumed (:311-313,332-347)
          char* p = nullptr; const unsigned long v = strtoul(t, &p, 0);
          if (p != t) return (uint32_t) v;
          if (!symbolOf) throw Error(DSR_E_KEY, "field '%s' is not a number and the transducer has no %s lexicon", t, which == 0 ? "state" : which == 1 ? "input" : "output");
          return symbolOf(which, t);
        };
        const uint32_t s1 = field(0, tok[0]);
        if (i == 1) addFinal(s1, 0.0f);
        else if (i == 2) { float c = 0.f; sscanf(tok[1], "%f", &c); addFinal(s1, c); }
        else if (i == 4 || i == 5) {
          const uint32_t s2 = field(0, tok[1]), in = field(1, tok[2]), out = field(2, tok[3]);
          if (s1 == s2 && noSelfLoops) continue;
          float c = 0.f; if (i == 5) sscanf(tok[4], "%f", &c);
          addArc(s1, s2, in, out, c, true);
        } else { free(line); throw Error(DSR_E_IO, "Transducer file is inconsistent."); }
      }
      free(line);
    }
  } catch (...) { fclose(fp); throw; }
  fclose(fp);
}

void WfstGraph::write(const char* file, bool binary, bool useSymbols) const
{
  if (!file || !*file) throw Error(DSR_E_IO, "Must specify a non-null file name for writing.");
  FILE* fp = fopen(file, binary ? "wb" : "w");
  if (!fp) throw Error(DSR_E_IO, "Could not open file %s", file);
  BEFile w{fp};
  auto writeArc = [&](const Arc& a) {                               // Edge::write :474-495; with symbols :499-516
    if (useSymbols) {
      // (the numeric writers below still serve the final-state lines and, with binary, the end marker: the reference mixes them the same way)
      if (!nameOf) throw Error(DSR_E_KEY, "write(useSymbols): the transducer has no lexica");
      const std::string in = nameOf(1, a.in), out = nameOf(2, a.out);
      if (!stateLexSize || stateLexSize() == 0) fprintf(fp, "%10d  %10d  %10s  %20s", (int) nodes[a.src].state, (int) nodes[a.dst].state, in.c_str(), out.c_str());
      else fprintf(fp, "%25s  %25s  %10s  %20s", nameOf(0, nodes[a.src].state).c_str(), nameOf(0, nodes[a.dst].state).c_str(), in.c_str(), out.c_str());
      if (std::fabs(a.cost) < 1.0E-04) fprintf(fp, "\n"); else fprintf(fp, "  %12g\n", (double) a.cost);      // MinimumCost (:497)
    } else if (binary) { w.w32(6); w.w32((int) nodes[a.src].state); w.w32((int) nodes[a.dst].state); w.w32((int) a.in); w.w32((int) a.out); w.wf(a.cost); w.w32(kEndMarker); }
    else { fprintf(fp, "%10d  %10d  %10d  %10d", (int) nodes[a.src].state, (int) nodes[a.dst].state, (int) a.in, (int) a.out);
      if (a.cost == 0.0) fprintf(fp, "\n"); else fprintf(fp, "  %12g\n", (double) a.cost); }
  };
  if (initial >= 0) for (int a = nodes[initial].firstArc; a >= 0; a = arcs[a].next) writeArc(arcs[a]);
  std::vector<std::pair<uint32_t, int>> ord;
  for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].inNodes) ord.push_back(std::make_pair(nodes[i].state, (int) i));
  std::sort(ord.begin(), ord.end());                                // std::map order
  for (size_t k = 0; k < ord.size(); k++) for (int a = nodes[ord[k].second].firstArc; a >= 0; a = arcs[a].next) writeArc(arcs[a]);
  ord.clear();
  for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].inFinal) ord.push_back(std::make_pair(nodes[i].state, (int) i));
  std::sort(ord.begin(), ord.end());
  for (size_t k = 0; k < ord.size(); k++) {
    const Node& nd = nodes[ord[k].second];
    for (int a = nd.firstArc; a >= 0; a = arcs[a].next) writeArc(arcs[a]);
    if (binary) { w.w32(3); w.w32((int) nd.state); w.wf(nd.cost); w.w32(kEndMarker); }          // Node::write :558-575
    else { if (nd.cost == 0.0) fprintf(fp, "%10d\n", (int) nd.state); else fprintf(fp, "%10d  %12g\n", (int) nd.state, (double) nd.cost); }
  }
  if (binary) w.w32(kEndMarker);
  fclose(fp);
}

WfstGraph::Csr WfstGraph::csr() const
{
  Csr c; const int n = (int) nodes.size();
  c.off.assign(n + 1, 0); c.csrOf.assign(arcs.size(), 0);
  int pos = 0;
  for (int i = 0; i < n; i++) { c.off[i] = pos; for (int a = nodes[i].firstArc; a >= 0; a = arcs[a].next) c.csrOf[a] = pos++; }
  c.off[n] = pos;
  c.dst.resize(pos); c.in.resize(pos); c.out.resize(pos); c.cost.resize(pos);
  for (size_t a = 0; a < arcs.size(); a++) { const int k = c.csrOf[a]; c.dst[k] = arcs[a].dst; c.in[k] = arcs[a].in; c.out[k] = arcs[a].out; c.cost[k] = arcs[a].cost; }
  return c;
}

WfstGraph::Tables WfstGraph::tables(const Csr& c, size_t maxRecords) const
{
  Tables t; const int n = (int) nodes.size();
  t.xoff.assign(n + 1, 0); t.eoff.assign(n + 1, 0);
  std::vector<int> path;               // current epsilon path (CSR arc ids)
  std::vector<uint8_t> onPath(n, 0);   // cycle detection
  struct Frame { int node; int arc; };
  std::vector<Frame> stack;
  for (int root = 0; root < n; root++) {
    t.xoff[root] = (int) t.xrec.size(); t.eoff[root] = (int) t.erec.size();
    // iterative depth-first walk mirroring _expandNode / _expandNodeToEnd
    stack.clear(); path.clear();
    stack.push_back(Frame{root, c.off[root]}); onPath[root] = 1;
    while (!stack.empty()) {
      Frame& f = stack.back();
      if (f.arc >= c.off[f.node + 1]) {
        onPath[f.node] = 0; stack.pop_back();
        if (!path.empty() && !stack.empty()) path.pop_back();
        continue;
      }
      const int a = f.arc++;
      if (c.in[a] == 0) {
        const int dstN = c.dst[a];
        if (onPath[dstN]) throw Error(DSR_E_CONSISTENCY, "epsilon cycle through state %u: the reference decoder recurses forever on it", nodes[dstN].state);
        path.push_back(a);
        if (nodes[dstN].final_) {       // _expandNodeToEnd places on entering a final state (decoder.h:1011-1012)
          ERec e{dstN, f.node, (int) t.path.size(), (int) path.size()};
          t.path.insert(t.path.end(), path.begin(), path.end());
          t.erec.push_back(e);
        }
        if (path.size() > 60000) throw Error(DSR_E_CONSISTENCY, "epsilon path longer than 60000 arcs");
        stack.push_back(Frame{dstN, c.off[dstN]}); onPath[dstN] = 1;
      } else {
        XRec x; x.dst = c.dst[a]; x.dist = (int) c.in[a] - 1; x.cost = c.cost[a];
        x.meta = (uint32_t) path.size() | (c.out[a] != 0 ? 0x10000u : 0u);
        t.xarc.push_back(a); t.xpathOff.push_back((int) t.path.size());
        if (!path.empty()) t.path.insert(t.path.end(), path.begin(), path.end());
        t.xrec.push_back(x);
      }
      if (t.xrec.size() + t.erec.size() > maxRecords || t.path.size() > 4 * maxRecords)
        throw Error(DSR_E_ALLOCATION, "epsilon expansion of the graph exceeds %zu records", maxRecords);
    }
  }
  t.xoff[n] = (int) t.xrec.size(); t.eoff[n] = (int) t.erec.size();
  if (t.path.empty()) t.path.push_back(0);
  return t;
}

}  // namespace dsr
