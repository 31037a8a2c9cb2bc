// csrc/wfst_graph.cpp -- see wfst_graph.h.
#include "wfst_graph.h"
#include <algorithm>
#include <climits>

namespace dsr {

static const int kEndMarker = 2147483647;        // WFSTFlyWeight::EndMarker (wfstFlyWeight.cc:32)

int WfstGraph::findNode(uint32_t state, bool create)
{
  if (initial >= 0 && nodes[initial].state == state) return initial;
  if (state >= nodeOf.size()) { size_t n = nodeOf.empty() ? 1024 : nodeOf.size(); while (n <= state) n *= 2; nodeOf.resize(n, -1); }
  if (nodeOf[state] >= 0) return nodeOf[state];
  if (!create) throw Error(DSR_E_KEY, "No state %u exists.", state);
  Node nd{state & 0x1FFFFFFFu, 0, 0.0f, -1, true, false};
  nodes.push_back(nd); nodeOf[state] = (int) nodes.size() - 1;
  return nodeOf[state];
}

void WfstGraph::addFinal(uint32_t state, float cost)
{
  if (state >= nodeOf.size()) { size_t n = nodeOf.empty() ? 1024 : nodeOf.size(); while (n <= state) n *= 2; nodeOf.resize(n, -1); }
  int id = nodeOf[state];
  if (id >= 0 && nodes[id].inFinal) throw Error(DSR_E_CONSISTENCY, "Automaton already has final node %d.", (int) state);
  if (id < 0) { Node nd{state & 0x1FFFFFFFu, 0, 0.0f, -1, false, false}; nodes.push_back(nd); id = (int) nodes.size() - 1; nodeOf[state] = id; }
  nodes[id].cost = cost; nodes[id].final_ = 1; nodes[id].inFinal = true; nodes[id].inNodes = false;
}

void WfstGraph::addArc(uint32_t s1, uint32_t s2, uint32_t in, uint32_t out, float cost, bool dropEpsSelf)
{
  int from;
  if (initial < 0) { Node nd{s1 & 0x1FFFFFFFu, 0, 0.0f, -1, false, false}; nodes.push_back(nd); initial = from = (int) nodes.size() - 1; }
  else from = findNode(s1, true);
  const int to = findNode(s2, true);
  if (dropEpsSelf && s1 == s2 && in == 0 && out == 0) return;
  Arc a{from, to, in, out, cost, nodes[from].firstArc};
  arcs.push_back(a); nodes[from].firstArc = (int) arcs.size() - 1;
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
  nodes.clear(); arcs.clear(); nodeOf.clear(); initial = -1;       // _clear()
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

void WfstGraph::write(const char* file, bool binary) const
{
  if (!file || !*file) throw Error(DSR_E_IO, "Must specify a non-null file name for writing.");
  FILE* fp = fopen(file, binary ? "wb" : "w");
  if (!fp) throw Error(DSR_E_IO, "Could not open file %s", file);
  BEFile w{fp};
  auto writeArc = [&](const Arc& a) {                               // Edge::write :474-495
    if (binary) { w.w32(6); w.w32((int) nodes[a.src].state); w.w32((int) nodes[a.dst].state); w.w32((int) a.in); w.w32((int) a.out); w.wf(a.cost); w.w32(kEndMarker); }
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
