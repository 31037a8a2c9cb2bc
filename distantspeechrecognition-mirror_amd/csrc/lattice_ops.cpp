// csrc/lattice_ops.cpp -- the operations of asr/lattice's Lattice on a host-side lattice (the decoder's, a gathered one or one read from a
// file): rescore / bestHypo (asr/lattice/lattice.cc:122-306), gammaProbs (:309-379), prune / pruneEdges (:648-713), purge (:776-841), write
// (:715-757), read (asr/fsm/fsm.h:3787-3873) and the 1-best writers writeCTM / writePhoneCTM / writeHypoHTK / writeWordConfs (:420-646).
// Host code on purpose: the reference runs these on its host lattice after the search, on hundreds to thousands of links.
//
// The reference keeps a node in one of three places -- the initial node, the vector _nodes (slot = index) or the map _final (key = index) -- and
// prune()/purge() renumber the nodes that remain in topological order without moving them: afterwards a node's slot or key is its OLD index, the
// number it prints under the new one.  Everything that walks "_allNodes()" or "_finis()" therefore walks slots/keys, and a second prune() clears
// _nodes[new index] (lattice.cc:674-675), whichever node lives there.  Restated as it is.
#include "lattice.h"
#include "lexicon.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

namespace dsr {
namespace {

const double kLogZero = 1.0E10;                               // fsm.h:62
const double kHuge = (double) 3.40282347e+38F;                // HUGE (math.h)

// negative log-probabilities added (fsm.cc:38-56)
double log_add(double ap, double bp)
{
  if (ap > kLogZero) throw Error(DSR_E_CONSISTENCY, "ap (%g) > LogZero (%g)", ap, kLogZero);
  if (bp > kLogZero) throw Error(DSR_E_CONSISTENCY, "bp (%g) > LogZero (%g)", bp, kLogZero);
  if (ap > bp) { const double t = ap; ap = bp; bp = t; }
  const double diff = ap - bp;
  const double z = exp(diff);
  if (std::isnan(z)) throw Error(DSR_E_CONSISTENCY, "ap - bp = %g returned NaN.", z);
  return ap - log(1.0 + z);
}

FILE* open_out(const char* file, const char* mode) {
  if (!file || !*file) return stdout;
  FILE* fp = fopen(file, mode);
  if (!fp) throw Error(DSR_E_IO, "Could not open file %s", file);
  return fp;
}
void close_out(FILE* fp) { if (fp != stdout) fclose(fp); else fflush(fp); }

}  // namespace
FILE* lat_open_out(const char* file, const char* mode) { return open_out(file, mode); }
void lat_close_out(FILE* fp) { close_out(fp); }

void LatticeData::ensure_ops()
{
  if (opsReady) return;
  const int n = (int) nodeFinal.size(); const size_t E = from.size();
  if (n == 0) throw Error(DSR_E_CONSISTENCY, "empty lattice object");
  for (size_t e = 0; e < E; e++) if (from[e] < 0 || from[e] >= n || to[e] < 0 || to[e] >= n) throw Error(DSR_E_INDEX, "edge %zu leaves the node range", e);
  if (index.size() != (size_t) n) { index.resize(n); for (int i = 0; i < n; i++) index[i] = i; }
  if (ecost.size() != E) ecost.assign(E, 0.0f);
  if (ncost.size() != (size_t) n) ncost.assign(n, 0.0f);
  if (gamma.size() != E) gamma.assign(E, 0.0);               // LatticeEdgeData: _gamma(0.0)
  fwd.assign(n, kLogZero); bwd.assign(n, kLogZero);          // LatticeNodeData: LogZero, LogZero, no token
  ntok.assign(n, -1); toks.clear(); color.assign(n, 0); success.assign(n, 0); sorted.clear();
  adj.assign(n, std::vector<int>());
  for (long e = (long) E - 1; e >= 0; e--) adj[from[e]].push_back((int) e);       // _addEdgeForce puts the new edge at the head (fsm.cc:540-544)
  if (slots.empty() && finals.empty()) {                     // as the decoder builds it: node k prints as k, lives in _nodes[k] or _final[k]
    slots.assign(n, -1);
    for (int i = 1; i < n; i++) { if (nodeFinal[i] == 1) finals.push_back(std::make_pair((unsigned) i, i)); else slots[i] = i; }
  }
  opsReady = true;
}

void LatticeData::topo_sort()
{
  if (!sorted.empty()) return;                               // lattice.cc:876
  // setColor(White): the initial node, the occupied slots, the final map (fsm.cc:428-441) -- a node that left all three keeps its colour
  color[0] = 0;
  for (size_t p = 0; p < slots.size(); p++) if (slots[p] >= 0) color[slots[p]] = 0;
  for (size_t k = 0; k < finals.size(); k++) color[finals[k].second] = 0;
  std::vector<int> order;                                    // finishing order; the reference pushes a finished node to the FRONT of its list
  struct Fr { int node; size_t k; };
  std::vector<Fr> st;
  auto enter = [&](int v) -> bool {
    if (color[v] == 2) return false;
    if (color[v] == 1) throw Error(DSR_E_CONSISTENCY, "Node %d is gray; graph is not acyclic.", index[v]);       // lattice.cc:862-864
    color[v] = 1; st.push_back(Fr{v, 0}); return true;
  };
  enter(0);
  while (!st.empty()) {
    Fr& fr = st.back();
    if (fr.k < adj[fr.node].size()) { const int v = to[adj[fr.node][fr.k++]]; enter(v); }
    else { color[fr.node] = 2; order.push_back(fr.node); st.pop_back(); }
  }
  sorted.assign(order.rbegin(), order.rend());
}

int LatticeData::best_token() const
{
  int best = -1; double bestScore = kHuge;
  for (size_t k = 0; k < finals.size(); k++) {
    const int t = ntok[finals[k].second];
    if (t < 0) continue;
    const float sc = toks[t].ac + toks[t].lm;                // Token::score(): a float sum
    if ((double) sc < bestScore) { best = t; bestScore = (double) sc; }
  }
  return best;
}

float LatticeData::rescore(double lmScale_, double lmPenalty_, double silPenalty_, unsigned silenceX_)
{
  ensure_ops();
  lmScale = lmScale_; lmPenalty = lmPenalty_; silPenalty = silPenalty_; silenceX = silenceX_;     // (_acScale stays what the last gammaProbs left)
  topo_sort();
  for (size_t i = 0; i < sorted.size(); i++) ntok[sorted[i]] = -1;                                // _clearTokens (:252-258)
  for (size_t i = 0; i < sorted.size(); i++) {                                                    // _expandNode (:136-171)
    const int node = sorted[i]; const int nt = ntok[node];
    double acScoreNode = 0.0, lmScoreNode = 0.0;
    if (nt >= 0) { acScoreNode = (double) toks[nt].ac; lmScoreNode = (double) toks[nt].lm; }
    for (size_t k = 0; k < adj[node].size(); k++) {
      const int e = adj[node][k];
      double acScore = acScoreNode + acScale * ac[e];
      double lmScore = lmScoreNode + lmScale * lm[e];
      if (out[e] != 0) lmScore += lmScale * lmPenalty;
      if (in[e] == silenceX && (nt < 0 || in[toks[nt].edge] != silenceX)) lmScore += lmScale * silPenalty;
      const double ttlScore = acScore + lmScore;
      const int xt = ntok[to[e]];
      if (xt < 0 || ttlScore < (double) (float) (toks[xt].ac + toks[xt].lm)) {
        Tok t; t.ac = (float) acScore; t.lm = (float) lmScore; t.edge = e; t.prev = nt;
        toks.push_back(t); ntok[to[e]] = (int) toks.size() - 1;
      }
    }
  }
  const int best = best_token();
  if (best < 0) throw Error(DSR_E_CONSISTENCY, "no token in a final node (the reference dereferences a null token here)");
  return toks[best].ac + toks[best].lm;
}

std::vector<unsigned> LatticeData::best_hypo(bool useInputSymbols) const
{
  if (!opsReady) throw Error(DSR_E_CONSISTENCY, "bestHypo before rescore");
  int t = best_token();
  if (t < 0) throw Error(DSR_E_CONSISTENCY, "no token in a final node (the reference dereferences a null token here)");
  std::vector<unsigned> rev; unsigned lastX = 0;
  do {
    if (useInputSymbols) { const unsigned inX = in[toks[t].edge]; if (inX != 0 && inX != lastX) { rev.push_back(inX); lastX = inX; } }
    else { const unsigned outX = out[toks[t].edge]; if (outX != 0) rev.push_back(outX); }
    t = toks[t].prev;
  } while (t >= 0);
  return std::vector<unsigned>(rev.rbegin(), rev.rend());
}

double LatticeData::gamma_probs(double acScale_, double lmScale_, double lmPenalty_, double silPenalty_, unsigned silenceX_)
{
  ensure_ops();
  acScale = acScale_; lmScale = lmScale_; lmPenalty = lmPenalty_; silPenalty = silPenalty_; silenceX = silenceX_;
  topo_sort();
  fwd[0] = 0.0; bwd[0] = kLogZero;
  for (size_t p = 0; p < slots.size(); p++) if (slots[p] >= 0) { fwd[slots[p]] = kLogZero; bwd[slots[p]] = kLogZero; }
  for (size_t k = 0; k < finals.size(); k++) { fwd[finals[k].second] = kLogZero; bwd[finals[k].second] = 0.0; }
  // the link score as all three passes form it (:181-191, 203-213, 228-238); the silence test looks at the node's rescoring token
  auto lm_of = [&](int node, int e) -> double {
    double lmScore = lmScale * lm[e];
    if (out[e] != 0) lmScore += lmScale * lmPenalty;
    const int nt = ntok[node];
    if (in[e] == silenceX && (nt < 0 || in[toks[nt].edge] != silenceX)) lmScore += lmScale * silPenalty;
    return lmScore;
  };
  for (size_t i = 0; i < sorted.size(); i++) {                                                    // _forwardProbs (:343-358)
    const int node = sorted[i]; const double scoreNode = fwd[node];
    for (size_t k = 0; k < adj[node].size(); k++) {
      const int e = adj[node][k];
      const double acScore = acScale * ac[e]; const double lmScore = lm_of(node, e);
      fwd[to[e]] = log_add(scoreNode + acScore + lmScore, fwd[to[e]]);
    }
  }
  latticeForwardProb = kLogZero;
  for (size_t k = 0; k < finals.size(); k++) latticeForwardProb = log_add(latticeForwardProb, fwd[finals[k].second]);
  for (size_t i = sorted.size(); i-- > 0;) {                                                      // _backwardProbs (:361-371)
    const int node = sorted[i];
    for (size_t k = 0; k < adj[node].size(); k++) {
      const int e = adj[node][k];
      const double acScore = acScale * ac[e]; const double lmScore = lm_of(node, e);
      const double ttlScore = bwd[to[e]] + acScore + lmScore;
      if (ttlScore >= kLogZero) continue;
      bwd[node] = log_add(ttlScore, bwd[node]);
    }
  }
  const double latticeBackwardProb = bwd[0];
  if ((fabs(latticeBackwardProb - latticeForwardProb) / latticeForwardProb) > 0.0001)
    throw Error(DSR_E_CONSISTENCY, "Lattice forward (%g) and backward probabilities (%g) are not equal.", latticeForwardProb, latticeBackwardProb);
  for (size_t i = 0; i < sorted.size(); i++) {                                                    // _gammaProbs (:374-378, 222-250)
    const int node = sorted[i];
    for (size_t k = 0; k < adj[node].size(); k++) {
      const int e = adj[node][k];
      const double acScore = acScale * ac[e]; const double lmScore = lm_of(node, e);
      double g = fwd[node] + acScore + lmScore + bwd[to[e]] - latticeForwardProb;
      if (g < 0.0) {
        if (g < -0.0001) throw Error(DSR_E_CONSISTENCY, "Warning: Neg. Log-Probability (%g) of edge %d --> %d is negative", g, index[from[e]], index[to[e]]);
        g = 0.0;
      }
      gamma[e] = g;
    }
  }
  return latticeForwardProb;
}

void LatticeData::prune(double threshold)
{
  ensure_ops();
  if (threshold < 0.0) throw Error(DSR_E_CONSISTENCY, "Lattice pruning threshold (%g) < 0.0.", threshold);
  auto remove_links = [&](int node) {                                                             // Node::_removeLinks (:949-965)
    std::vector<int>& a = adj[node]; size_t w = 0;
    for (size_t k = 0; k < a.size(); k++) if (!(gamma[a[k]] > threshold)) a[w++] = a[k];
    a.resize(w);
  };
  remove_links(0);
  for (size_t p = 0; p < slots.size(); p++) if (slots[p] >= 0) remove_links(slots[p]);           // (the final nodes keep their links)
  sorted.clear(); topo_sort();
  std::vector<int> gone;                                                                          // by INDEX, as the reference collects them
  for (size_t p = 0; p < slots.size(); p++) if (slots[p] >= 0 && color[slots[p]] == 0) gone.push_back(index[slots[p]]);
  for (size_t i = 0; i < gone.size(); i++) {
    if ((size_t) gone[i] >= slots.size()) throw Error(DSR_E_INDEX, "prune: _nodes[%d] lies past the vector (the reference writes there unchecked)", gone[i]);
    slots[gone[i]] = -1;
  }
  size_t w = 0;
  for (size_t k = 0; k < finals.size(); k++) if (color[finals[k].second] != 0) finals[w++] = finals[k];
  finals.resize(w);
  for (size_t i = 0; i < sorted.size(); i++) index[sorted[i]] = (int) i;
}

void LatticeData::prune_edges(unsigned edgesN)
{
  ensure_ops();
  std::vector<double> scores;                                                                     // EdgeIterator (:984-1002): initial node + _nodes, not the finals
  for (size_t k = 0; k < adj[0].size(); k++) scores.push_back(gamma[adj[0][k]]);
  for (size_t p = 0; p < slots.size(); p++) if (slots[p] >= 0) for (size_t k = 0; k < adj[slots[p]].size(); k++) scores.push_back(gamma[adj[slots[p]][k]]);
  if ((size_t) edgesN >= scores.size()) return;
  std::sort(scores.begin(), scores.end());
  prune(scores[edgesN]);
}

void LatticeData::purge()
{
  ensure_ops();
  color[0] = 0; success[0] = 0;                                                                   // setColor(White); _setSuccess(false)
  for (size_t p = 0; p < slots.size(); p++) if (slots[p] >= 0) { color[slots[p]] = 0; success[slots[p]] = 0; }
  for (size_t k = 0; k < finals.size(); k++) { color[finals[k].second] = 0; success[finals[k].second] = 0; }
  sorted.clear();
  // _purgeNode (:776-800), depth first with an explicit stack: success = final || any successor successful; successful nodes to the front
  std::vector<int> order; struct Fr { int node; size_t k; int succ; }; std::vector<Fr> st;
  auto enter = [&](int v) { color[v] = 1; st.push_back(Fr{v, 0, nodeFinal[v] == 1 ? 1 : 0}); };
  enter(0);
  while (!st.empty()) {
    Fr& fr = st.back();
    if (fr.k < adj[fr.node].size()) {
      const int v = to[adj[fr.node][fr.k++]];
      if (color[v] == 2) { if (success[v]) fr.succ = 1; }
      else if (color[v] == 1) throw Error(DSR_E_CONSISTENCY, "Node %d is gray; graph is not acyclic.", index[v]);
      else enter(v);
    } else {
      const int node = fr.node, s = fr.succ; st.pop_back();
      success[node] = s; color[node] = 2;
      if (s) order.push_back(node);
      if (!st.empty() && s) st.back().succ = 1;
    }
  }
  sorted.assign(order.rbegin(), order.rend());
  std::vector<int> gone;                                                                          // _removeUnsuccessful (:802-828)
  for (size_t p = 0; p < slots.size(); p++) if (slots[p] >= 0 && !success[slots[p]]) gone.push_back(index[slots[p]]);
  for (size_t i = 0; i < gone.size(); i++) {
    if ((size_t) gone[i] >= slots.size()) throw Error(DSR_E_INDEX, "purge: _nodes[%d] lies past the vector (the reference writes there unchecked)", gone[i]);
    slots[gone[i]] = -1;
  }
  size_t w = 0;
  for (size_t k = 0; k < finals.size(); k++) if (success[finals[k].second]) finals[w++] = finals[k];
  finals.resize(w);
  for (size_t i = 0; i < sorted.size(); i++) index[sorted[i]] = (int) i;
}

// Lattice::write(fileName, useSymbols = false, writeData) (lattice.cc:715-757): the links of the non-final nodes in topological order, then per
// final node (key order) its links and its node line; formats fsm.cc:553-559, 1171-1178 and lattice.h:151-154
void LatticeData::write(const char* file, bool writeData)
{
  ensure_ops();
  topo_sort();
  FILE* fp = open_out(file, "w");
  auto wedge = [&](int e) {
    fprintf(fp, "%10d  %10d  %10d  %10d", index[from[e]], index[to[e]], (int) in[e], (int) out[e]);
    if (fabs(ecost[e]) < 1.0E-04) fprintf(fp, "\n"); else fprintf(fp, "  %12g\n", ecost[e]);
    if (writeData) fprintf(fp, "%4d  %4d  %8.4f  %8.4f  %8.4f\n", start[e], end[e], ac[e], lm[e], gamma[e]);
  };
  for (size_t i = 0; i < sorted.size(); i++) { const int nd = sorted[i]; if (nodeFinal[nd] == 1) continue; for (size_t k = 0; k < adj[nd].size(); k++) wedge(adj[nd][k]); }
  for (size_t f = 0; f < finals.size(); f++) {
    const int nd = finals[f].second;
    for (size_t k = 0; k < adj[nd].size(); k++) wedge(adj[nd][k]);
    if (ncost[nd] == 0.0f) fprintf(fp, "%10d\n", index[nd]); else fprintf(fp, "%10d  %12g\n", index[nd], ncost[nd]);
  }
  close_out(fp);
}

// WFST<LatticeNodeData, LatticeEdgeData>::read (fsm.h:3787-3873): one line per link "from to input output [cost]" or final node "state [cost]";
// with readData a link is followed by "start end ac lm gamma", read with fscanf (which also eats the white space after it).  The first state
// named becomes the initial node; symbols that are not numbers (strtoul, base 0) are looked up in the lexica.
LatticeData LatticeData::read(const char* file, bool noSelfLoops, bool readData, dsr_lexicon* inlex, dsr_lexicon* outlex)
{
  if (!file || !*file) throw Error(DSR_E_IO, "File name is null.");
  FILE* fp = fopen(file, "r");
  if (!fp) throw Error(DSR_E_IO, "Could not open file %s", file);
  std::string buf; { char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.append(tmp, n); }
  fclose(fp);
  LatticeData L; L.finalStatesN = 0;
  bool haveInitial = false;
  auto new_node = [&](unsigned state) -> int { L.nodeFinal.push_back(0); L.index.push_back((int) state); L.ncost.push_back(0.0f); return (int) L.nodeFinal.size() - 1; };
  auto final_at = [&](unsigned state) -> int { for (size_t k = 0; k < L.finals.size(); k++) if (L.finals[k].first == state) return (int) k; return -1; };
  auto find = [&](unsigned state) -> int {                                                        // _find(state, create = true) (fsm.cc:153-178)
    if (!haveInitial) { haveInitial = true; return new_node(state); }                             // node 0
    if ((unsigned) L.index[0] == state) return 0;
    const int k = final_at(state); if (k >= 0) return L.finals[k].second;
    if (state < L.slots.size() && L.slots[state] >= 0) return L.slots[state];
    if (state >= L.slots.size()) L.slots.resize((size_t) state + 1, -1);
    const int nd = new_node(state); L.slots[state] = nd; return nd;
  };
  auto add_final = [&](unsigned state, float cost) {                                              // _addFinal (fsm.cc:122-138)
    if (final_at(state) >= 0) throw Error(DSR_E_CONSISTENCY, "Automaton already has final node %u.", state);
    int nd;
    if (state >= L.slots.size() || L.slots[state] < 0) nd = new_node(state);
    else { nd = L.slots[state]; L.slots[state] = -1; }
    L.nodeFinal[nd] = 1; L.ncost[nd] = cost;
    size_t pos = 0; while (pos < L.finals.size() && L.finals[pos].first < state) pos++;
    L.finals.insert(L.finals.begin() + pos, std::make_pair(state, nd));
  };
  auto symbol = [&](const std::string& t, dsr_lexicon* lex) -> unsigned {
    char* p = nullptr; const unsigned long v = strtoul(t.c_str(), &p, 0);
    if (p != t.c_str()) return (unsigned) v;
    if (!lex) throw Error(DSR_E_KEY, "symbol %s and no lexicon to look it up in", t.c_str());
    return lex->index(t);
  };
  size_t pos = 0;
  while (pos < buf.size()) {
    size_t eol = buf.find('\n', pos); if (eol == std::string::npos) eol = buf.size();
    const std::string line = buf.substr(pos, eol - pos); pos = (eol < buf.size()) ? eol + 1 : eol;
    std::vector<std::string> tok; { size_t a = 0; while (tok.size() < 6) { a = line.find_first_not_of(" \t", a); if (a == std::string::npos) break; size_t b = line.find_first_of(" \t", a); if (b == std::string::npos) b = line.size(); tok.push_back(line.substr(a, b - a)); a = b; } }
    if (tok.empty()) throw Error(DSR_E_PARSE, "Transducer file %s has an empty line (the reference reads a null token there)", file);
    const size_t i = tok.size() < 5 ? tok.size() : 5;
    unsigned s1 = 0; sscanf(tok[0].c_str(), "%u", &s1);
    if (i == 1 || i == 2) {
      float cost = 0.0f; if (i == 2) sscanf(tok[1].c_str(), "%f", &cost);
      if (!haveInitial) throw Error(DSR_E_CONSISTENCY, "final node %u before any link (the reference has no initial node yet)", s1);
      add_final(s1, cost);                                                                        // (LatticeNodeData::read reads nothing, lattice.h:98-100)
    } else if (i == 4 || i == 5) {
      unsigned s2 = 0; sscanf(tok[1].c_str(), "%u", &s2);
      if (s1 == s2 && noSelfLoops) continue;                                                      // (a data line that follows is then read as a link, as in the reference)
      const int fromN = find(s1); const int toN = find(s2);
      const unsigned input = symbol(tok[2], inlex), output = symbol(tok[3], outlex);
      if (s1 == s2 && input == 0) continue;
      float cost = 0.0f; if (i == 5) sscanf(tok[4].c_str(), "%f", &cost);
      int st = -1, en = -1; double a = 0.0, l = 0.0, g = 0.0;                                      // LatticeEdgeData(): -1, -1, 0, 0, 0
      if (readData) {
        int used = 0; const int nmatch = sscanf(buf.c_str() + pos, "%d %d %lf %lf %lf%n", &st, &en, &a, &l, &g, &used);
        if (nmatch != 5) throw Error(DSR_E_IO, "Only matched %d elements.", nmatch < 0 ? 0 : nmatch);
        pos += (size_t) used; while (pos < buf.size() && isspace((unsigned char) buf[pos])) pos++;
      }
      L.from.push_back(fromN); L.to.push_back(toN); L.in.push_back(input); L.out.push_back(output); L.start.push_back(st); L.end.push_back(en);
      L.ac.push_back(a); L.lm.push_back(l); L.gamma.push_back(g); L.ecost.push_back(cost);
    } else throw Error(DSR_E_IO, "Transducer file %s is inconsistent.", file);
  }
  if (!haveInitial) { L.nodeFinal.assign(1, 0); L.index.assign(1, 0); L.ncost.assign(1, 0.0f); }  // Lattice(): _initial = _newNode(0) stays after an empty file?  _clear() drops it; an empty lattice either way
  if (L.slots.empty() && L.finals.empty()) L.slots.assign(1, -1);                                 // (keeps ensure_ops from applying the decoder's layout)
  L.finalStatesN = (int) L.finals.size();
  return L;
}

}  // namespace dsr

// ------------------------------------------------------------------------------------------------------------------ C-ABI
using dsr::Error; using dsr::guard;
static FILE* open_out(const char* file, const char* mode) { return dsr::lat_open_out(file, mode); }
static void close_out(FILE* fp) { dsr::lat_close_out(fp); }

dsr_status dsr_lattice_read(const char* fileName, int noSelfLoops, int readData, dsr_lexicon* inlex, dsr_lexicon* outlex, dsr_lattice** out)
{
  return guard([&] {
    if (!fileName || !out) throw Error(DSR_E_PARAMETER, "null argument");
    dsr_lattice* L = new dsr_lattice();
    try { static_cast<dsr::LatticeData&>(*L) = dsr::LatticeData::read(fileName, noSelfLoops != 0, readData != 0, inlex, outlex); } catch (...) { delete L; throw; }
    *out = L;
  });
}
dsr_status dsr_lattice_rescore(dsr_lattice* L, double lmScale, double lmPenalty, double silPenalty, unsigned silenceX, float* score)
{ return guard([&] { if (!L) throw Error(DSR_E_PARAMETER, "null argument"); const float s = L->rescore(lmScale, lmPenalty, silPenalty, silenceX); if (score) *score = s; }); }
dsr_status dsr_lattice_best_hypo(const dsr_lattice* L, int useInputSymbols, uint32_t* symbols, int cap, int* n)
{
  return guard([&] {
    if (!L || !n) throw Error(DSR_E_PARAMETER, "null argument");
    const std::vector<unsigned> h = L->best_hypo(useInputSymbols != 0);
    *n = (int) h.size();
    if (symbols) { if (cap < (int) h.size()) throw Error(DSR_E_DIMENSION, "hypothesis of %zu symbols, room for %d", h.size(), cap); for (size_t i = 0; i < h.size(); i++) symbols[i] = h[i]; }
  });
}
dsr_status dsr_lattice_gamma_probs(dsr_lattice* L, double acScale, double lmScale, double lmPenalty, double silPenalty, unsigned silenceX, double* logProb)
{ return guard([&] { if (!L) throw Error(DSR_E_PARAMETER, "null argument"); const double p = L->gamma_probs(acScale, lmScale, lmPenalty, silPenalty, silenceX); if (logProb) *logProb = p; }); }
dsr_status dsr_lattice_prune(dsr_lattice* L, double threshold) { return guard([&] { if (!L) throw Error(DSR_E_PARAMETER, "null argument"); L->prune(threshold); }); }
dsr_status dsr_lattice_prune_edges(dsr_lattice* L, unsigned edgesN) { return guard([&] { if (!L) throw Error(DSR_E_PARAMETER, "null argument"); L->prune_edges(edgesN); }); }
dsr_status dsr_lattice_purge(dsr_lattice* L) { return guard([&] { if (!L) throw Error(DSR_E_PARAMETER, "null argument"); L->purge(); }); }
dsr_status dsr_lattice_get_state(dsr_lattice* L, double* gamma, int32_t* edgeLive, int32_t* nodeIndex, int32_t* nodeLive, double* forwardProb, double* backwardProb)
{
  return guard([&] {
    if (!L) throw Error(DSR_E_PARAMETER, "null argument");
    L->ensure_ops();
    const size_t E = L->from.size(), n = L->nodeFinal.size();
    if (gamma) for (size_t e = 0; e < E; e++) gamma[e] = L->gamma[e];
    if (edgeLive) { for (size_t e = 0; e < E; e++) edgeLive[e] = 0; for (size_t v = 0; v < n; v++) for (size_t k = 0; k < L->adj[v].size(); k++) edgeLive[L->adj[v][k]] = 1; }
    if (nodeIndex) for (size_t v = 0; v < n; v++) nodeIndex[v] = L->index[v];
    if (nodeLive) {
      for (size_t v = 0; v < n; v++) nodeLive[v] = 0;
      nodeLive[0] = 1;
      for (size_t p = 0; p < L->slots.size(); p++) if (L->slots[p] >= 0) nodeLive[L->slots[p]] = 1;
      for (size_t k = 0; k < L->finals.size(); k++) nodeLive[L->finals[k].second] = 1;
    }
    if (forwardProb) for (size_t v = 0; v < n; v++) forwardProb[v] = L->fwd[v];
    if (backwardProb) for (size_t v = 0; v < n; v++) backwardProb[v] = L->bwd[v];
  });
}

namespace {
// the 1-best chain, last link first, as the writers walk it
struct Best { const dsr::LatticeData& L; int t; };
void need(const dsr_lattice* L, const dsr_lexicon* lex) { if (!L || !lex) throw Error(DSR_E_PARAMETER, "null argument"); if (!L->opsReady) throw Error(DSR_E_CONSISTENCY, "no rescoring tokens: call rescore first"); }
int best_or_throw(const dsr_lattice* L) { const int t = L->best_token(); if (t < 0) throw Error(DSR_E_CONSISTENCY, "no token in a final node (the reference dereferences a null token here)"); return t; }
const char* nz(const char* s) { return s ? s : ""; }
}

// Lattice::writeCTM (lattice.cc:420-477): ";; utt cfrom score", then a row per output symbol of the 1-best chain ("a:b" entries split in two
// halves as the shipped loop does it); rows whose word is endMarker are skipped; the file is appended to
dsr_status dsr_lattice_write_ctm(const dsr_lattice* L, const dsr_lexicon* outlex, const char* conv, const char* channel, const char* spk, const char* utt,
                                 double cfrom, double score, const char* fileName, double frameInterval, const char* endMarker)
{
  return guard([&] {
    need(L, outlex); (void) spk;
    int t = best_or_throw(L);
    std::vector<std::string> words; std::vector<double> starts, durations, scores;
    int endX = L->end[L->toks[t].edge]; double wscore = (double) L->toks[t].ac;
    do {
      const dsr::LatticeData::Tok& tk = L->toks[t];
      const unsigned outX = L->out[tk.edge];
      if (outX != 0) {
        const int startX = L->start[tk.edge];
        double beg = cfrom + startX * frameInterval; double len = (endX - startX) * frameInterval;
        std::string entry = outlex->symbol(outX);
        std::string::size_type colon;
        do {
          colon = entry.find(":");
          std::string word = entry;
          if (colon != std::string::npos) { word = entry.substr(colon + 1); len /= 2; beg += len; }
          words.push_back(word); starts.push_back(beg); durations.push_back(len);
          const double oscore = (tk.prev < 0) ? 0.0 : (double) L->toks[tk.prev].ac;
          scores.push_back(wscore - oscore);
          endX = startX; wscore = oscore;
          if (colon != std::string::npos) { entry = entry.substr(0, colon); beg = cfrom + startX * frameInterval; }
        } while (colon != std::string::npos);
      }
      t = tk.prev;
    } while (t >= 0);
    FILE* fp = open_out(fileName, "a");
    fprintf(fp, ";; %s %10.4f %10.4f\n", nz(utt), cfrom, score);
    for (int i = (int) words.size() - 1; i >= 0; i--) {
      if (words[i] == nz(endMarker)) continue;
      fprintf(fp, "%s %s %7.2f %7.2f %-20s %7.2f\n", nz(conv), nz(channel), starts[i], durations[i], words[i].c_str(), scores[i]);
    }
    close_out(fp);
  });
}

// Lattice::writePhoneCTM (lattice.cc:479-537): a row per link of the chain (the shipped test "phoneX != 0 || phoneX != lastPhoneX" only drops an
// epsilon link that follows an epsilon link)
dsr_status dsr_lattice_write_phone_ctm(const dsr_lattice* L, const dsr_lexicon* inlex, const char* conv, const char* channel, const char* spk, const char* utt,
                                       double cfrom, double score, const char* fileName, double frameInterval, const char* endMarker)
{
  return guard([&] {
    need(L, inlex); (void) spk;
    int t = best_or_throw(L);
    std::vector<std::string> phones; std::vector<double> starts, durations, scores;
    int endX = L->end[L->toks[t].edge]; double wscore = (double) L->toks[t].ac; unsigned lastPhoneX = 0;
    do {
      const dsr::LatticeData::Tok& tk = L->toks[t];
      const unsigned phoneX = L->in[tk.edge];
      if (phoneX != 0 || phoneX != lastPhoneX) {
        const int startX = L->start[tk.edge];
        phones.push_back(inlex->symbol(phoneX)); starts.push_back(cfrom + startX * frameInterval); durations.push_back((endX - startX) * frameInterval);
        const double oscore = (tk.prev < 0) ? 0.0 : (double) L->toks[tk.prev].ac;
        scores.push_back(wscore - oscore);
        endX = startX; lastPhoneX = phoneX; wscore = oscore;
      }
      t = tk.prev;
    } while (t >= 0);
    FILE* fp = open_out(fileName, "a");
    fprintf(fp, ";; %s %10.4f %10.4f\n", nz(utt), cfrom, score);
    for (int i = (int) phones.size() - 1; i >= 0; i--) {
      if (phones[i] == nz(endMarker)) continue;
      fprintf(fp, "%s %s %7.2f %7.2f %-20s %7.2f\n", nz(conv), nz(channel), starts[i], durations[i], phones[i].c_str(), scores[i]);
    }
    close_out(fp);
  });
}

// Lattice::writeHypoHTK (lattice.cc:539-601): "utt.rec", a line per word (flag bit 0: times in 100-ns units, bit 1: score), "."
dsr_status dsr_lattice_write_hypo_htk(const dsr_lattice* L, const dsr_lexicon* outlex, const char* conv, const char* channel, const char* spk, const char* utt,
                                      double cfrom, double score, const char* fileName, int flag, double frameInterval, const char* endMarker)
{
  return guard([&] {
    need(L, outlex); (void) conv; (void) channel; (void) spk; (void) score;
    int t = best_or_throw(L);
    std::vector<std::string> words; std::vector<double> starts, durations, scores;
    int endX = L->end[L->toks[t].edge]; double wscore = (double) L->toks[t].ac;
    do {
      const dsr::LatticeData::Tok& tk = L->toks[t];
      const unsigned outX = L->out[tk.edge];
      if (outX != 0) {
        const int startX = L->start[tk.edge];
        words.push_back(outlex->symbol(outX)); starts.push_back(cfrom + startX * frameInterval); durations.push_back((endX - startX + 1) * frameInterval);
        const double oscore = (tk.prev < 0) ? 0.0 : (double) L->toks[tk.prev].ac;
        scores.push_back(wscore - oscore);
        endX = startX - 1; wscore = oscore;
      }
      t = tk.prev;
    } while (t >= 0);
    FILE* fp = open_out(fileName, "a");
    fprintf(fp, "\"%s.rec\"\n", nz(utt));
    for (int i = (int) words.size() - 1; i >= 0; i--) {
      if (words[i] == nz(endMarker)) continue;
      if (flag & 0x01) fprintf(fp, "%lld %lld ", (long long) (starts[i] * 10e7), (long long) ((starts[i] + durations[i]) * 10e7));
      fprintf(fp, "%s", words[i].c_str());
      if (flag & 0x02) fprintf(fp, " %f", scores[i]);
      fprintf(fp, "\n");
    }
    fprintf(fp, ".\n");
    close_out(fp);
  });
}

// Lattice::writeWordConfs (lattice.cc:603-646): "uttId { {word} conf} ..." with conf = exp(-gamma) of the word's link, clipped as the reference clips it
dsr_status dsr_lattice_write_word_confs(const dsr_lattice* L, const dsr_lexicon* outlex, const char* fileName, const char* uttId, const char* endMarker)
{
  return guard([&] {
    need(L, outlex);
    int t = best_or_throw(L);
    std::vector<std::string> words; std::vector<double> gammas;
    do {
      const dsr::LatticeData::Tok& tk = L->toks[t];
      const unsigned outX = L->out[tk.edge];
      if (outX != 0) { words.push_back(outlex->symbol(outX)); gammas.push_back(L->gamma[tk.edge]); }
      t = tk.prev;
    } while (t >= 0);
    std::string output(nz(uttId)); char buffer[1200];
    for (int i = (int) words.size() - 1; i >= 0; i--) {
      double g = exp(-gammas[i]);
      if (g < 1.0E-04) g = 0.0; else if (g > 1.0) g = 1.0;
      if (words[i] == nz(endMarker)) continue;
      snprintf(buffer, sizeof(buffer), " { {%s} %8.6f}", words[i].c_str(), g);
      output += buffer;
    }
    FILE* fp = open_out(fileName, "a");
    fprintf(fp, "%s\n", output.c_str());
    close_out(fp);
  });
}
