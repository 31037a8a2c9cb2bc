// csrc/lattice.cpp -- see lattice.h.
//
// The decoder kernel logs, per frame and in arrival order, every placement {ac, lm, expansion record, parent back pointer, unrounded total}.
// From that log this file rebuilds exactly the token graph the reference holds at the end of decode():
//   * tokens: one per placement, plus the intermediate epsilon tokens of its expansion path (decoder.h:979-983, 992-1015), whose float scores are
//     recomputed hop by hop with the roundings of Token's constructor (lattice.h:41-44);
//   * prev: the token before on the path (the parent's winning placement for the first hop);
//   * worse: _placeOnList's chains (decoder.h:531-541), an order-dependent function of the arrivals at one state: replayed literally in slot order;
// and then runs the reference's trace (lattice(), _majorTrace, _minorTrace, _findLNode) with an explicit stack instead of its recursion.
#include "lattice.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <unordered_map>

namespace dsr {

namespace {

static constexpr uint32_t kNone = 0xFFFFFFFFu, kSelf = 0x7FFFFFFEu;
typedef long long Tok;                                     // placement index << 16 | (hop + 1); hop field 0 = the placed token itself; -1 = null

struct Builder {
  const LatInput& I; LatticeData& L;
  std::vector<int> arcSrc; std::vector<long> worse; std::vector<int> bucketOf;
  std::map<std::pair<unsigned, int>, int> lnodes; int lnStateIndices = 0;
  std::vector<std::vector<int> > edgesOf;                  // per lattice node, creation order

  explicit Builder(const LatInput& in, LatticeData& out) : I(in), L(out) {}
  static float f(uint32_t b) { float v; memcpy(&v, &b, 4); return v; }
  long nPlace() const { return I.frameOff[I.T + 1]; }
  bool endBucket(long p) const { return bucketOf[p] == I.T; }
  int frameOf(long p) const { return endBucket(p) ? I.T - 1 : bucketOf[p]; }     // _expandToEnd runs after _frameX-- (decoder.h:709-711)

  Tok parentTok(long p) const { const uint32_t bp = I.place[p].prevBp; return bp == kNone ? -1 : ((Tok) I.arenaLat[bp] << 16); }
  // the path of a placement: npath epsilon arcs, then (not for the end expansion) the emitting arc
  void pathOf(long p, const int*& arcs, int& nHop, int& lastArc) const {
    const uint32_t rec = I.place[p].rec;
    if (endBucket(p)) { const ERec& e = I.tab->erec[rec & 0x3FFFFFFFu]; arcs = &I.tab->path[e.pathOff]; nHop = e.pathLen - 1; lastArc = arcs[e.pathLen - 1]; }
    else { const uint32_t r = rec & 0x3FFFFFFFu; arcs = &I.tab->path[I.tab->xpathOff[r]]; nHop = (int) (I.tab->xrec[r].meta & 0xFFFFu); lastArc = I.tab->xarc[r]; }
  }
  bool isSelf(long p) const { return endBucket(p) && I.place[p].rec == kSelf; }
  int arcOf(Tok t) const {
    const long p = (long) (t >> 16); const int h = (int) (t & 0xFFFF);
    if (isSelf(p)) return arcOf(parentTok(p));                                    // _placeOnList(tok->edge(), ...) (decoder.h:506-509)
    const int* arcs; int nHop, last; pathOf(p, arcs, nHop, last);
    return h == 0 ? last : arcs[h - 1];
  }
  Tok prevOf(Tok t) const {
    const long p = (long) (t >> 16); const int h = (int) (t & 0xFFFF);
    if (isSelf(p)) return prevOf(parentTok(p));                                   // ... tok->prev())
    if (h == 0) { const int* arcs; int nHop, last; pathOf(p, arcs, nHop, last); return nHop > 0 ? (((Tok) p << 16) | (Tok) nHop) : parentTok(p); }
    return h > 1 ? (((Tok) p << 16) | (Tok) (h - 1)) : parentTok(p);
  }
  int frameOfTok(Tok t) const { return frameOf((long) (t >> 16)); }
  // float scores of a token (acScore, lmScore)
  void scoresOf(Tok t, float& ac, float& lm) const {
    const long p = (long) (t >> 16); const int h = (int) (t & 0xFFFF);
    if (h == 0) { ac = f(I.place[p].ac); lm = f(I.place[p].lm); return; }
    // intermediate epsilon token h (1-based) of placement p: walk the hops from the parent
    const Tok par = parentTok(p); float pac = 0.0f, plm = 0.0f; bool prevNull = par < 0; uint32_t prevIn = I.silenceX + 1u;
    if (!prevNull) { scoresOf(par, pac, plm); prevIn = I.csr->in[arcOf(par)]; }
    const int* arcs; int nHop, last; pathOf(p, arcs, nHop, last);
    double lmNode = (double) plm;
    for (int k = 0; k < h; k++) {
      const int a = arcs[k];
      double l = lmNode + I.lmScale * (double) I.csr->cost[a];
      if (I.csr->out[a] != 0) l += (I.lmScale * I.lmPenalty);
      if (endBucket(p)) { if (0u == I.silenceX && prevIn != I.silenceX) l += (I.lmScale * I.silPenalty); }            // decoder.h:1004-1006
      else if (0u == I.silenceX && (prevNull || prevIn != I.silenceX)) l += (I.lmScale * I.silPenalty);               // decoder.h:975-977
      lmNode = (double) (float) l; prevIn = 0u; prevNull = false;
    }
    ac = pac; lm = (float) lmNode;
  }
  int dstNodeOf(long p) const {
    if (isSelf(p)) return I.csr->dst[arcOf(parentTok(p))];
    const uint32_t rec = I.place[p].rec;
    return endBucket(p) ? I.tab->erec[rec & 0x3FFFFFFFu].dst : I.tab->xrec[rec & 0x3FFFFFFFu].dst;
  }

  void prepare() {
    const int nNodes = (int) I.graph->nodes.size();
    arcSrc.assign(I.csr->dst.size(), 0);
    for (int n = 0; n < nNodes; n++) for (int a = I.csr->off[n]; a < I.csr->off[n + 1]; a++) arcSrc[a] = n;
    const long N = nPlace();
    bucketOf.assign((size_t) N, 0); worse.assign((size_t) N, -1);
    for (int b = 0; b <= I.T; b++) for (long p = I.frameOff[b]; p < I.frameOff[b + 1]; p++) bucketOf[p] = b;
    // 'worse' chains: arrivals of one destination state, in slot order (decoder.h:519-541)
    std::unordered_map<int, long> head;
    for (int b = 0; b <= I.T; b++) {
      head.clear();
      for (long p = I.frameOff[b]; p < I.frameOff[b + 1]; p++) {
        const int dst = dstNodeOf(p);
        auto it = head.find(dst);
        if (it == head.end()) { head.emplace(dst, p); continue; }
        const long inc = it->second;
        const float incScore = f(I.place[inc].ac) + f(I.place[inc].lm);            // Token::score(): float sum (lattice.h:53)
        if (I.ttl[p] < (double) incScore) { worse[p] = inc; it->second = p; }
        else { worse[p] = worse[inc]; worse[inc] = p; }
      }
      // consistency: the kernel's winner of every written state is the head of its chain
    }
  }

  int addNode(bool final_) { const int idx = ++lnStateIndices; if ((int) L.nodeFinal.size() <= idx) { L.nodeFinal.resize(idx + 1, 0); edgesOf.resize(idx + 1); } L.nodeFinal[idx] = final_ ? 1 : 0; return idx; }
  void addEdge(int from, int to, uint32_t in, uint32_t out, int s, int e, double ac, double lm) {
    L.from.push_back(from); L.to.push_back(to); L.in.push_back(in); L.out.push_back(out); L.start.push_back(s); L.end.push_back(e); L.ac.push_back(ac); L.lm.push_back(lm);
  }
  unsigned stateIndex(int node) const { return I.graph->nodes[node].state; }

  // _minorTrace (decoder.h:873-931); returns the token to continue with (or -1) and its lattice node key
  bool minorTrace(Tok endTok, std::pair<unsigned, int> prevKey, Tok& contTok, std::pair<unsigned, int>& contKey) {
    Tok tok = endTok; const int endFrame = frameOfTok(endTok);
    uint32_t currentInput = I.csr->in[arcOf(tok)], currentOutput = I.csr->out[arcOf(tok)];
    for (;;) {
      const Tok pv = prevOf(tok); if (pv < 0) break;
      const int pa = arcOf(pv); const uint32_t pin = I.csr->in[pa], pout = I.csr->out[pa];
      if (pout != 0 && currentOutput != 0) break;
      if (pin != 0) { if (currentInput != 0 && pin != currentInput) break; currentInput = pin; }
      if (pout != 0) { if (currentOutput != 0) break; currentOutput = pout; }
      tok = pv;
    }
    const int begFrame = frameOfTok(tok); const Tok prevTok = prevOf(tok);
    const int endNode = lnodes.at(prevKey);
    float eac, elm; scoresOf(endTok, eac, elm);
    double acScore = (double) eac, lmScore = (double) elm;
    int begNode; bool create = true; std::pair<unsigned, int> lnode(0u, 0);
    if (prevTok < 0) begNode = 0;
    else {
      float pac, plm; scoresOf(prevTok, pac, plm);
      acScore -= (double) pac; lmScore -= (double) plm;
      lnode = std::make_pair(stateIndex(arcSrc[arcOf(tok)]), begFrame);
      auto it = lnodes.find(lnode);
      if (it != lnodes.end()) { begNode = it->second; create = false; }
      else { begNode = addNode(false); lnodes.emplace(lnode, begNode); }
    }
    if (currentInput == I.silenceX) lmScore -= (I.lmScale * I.silPenalty);
    if (currentOutput != 0) lmScore -= (I.lmScale * I.lmPenalty);
    lmScore /= I.lmScale;
    addEdge(begNode, endNode, currentInput, currentOutput, begFrame, endFrame, acScore, lmScore);
    if (prevTok >= 0 && create) { contTok = prevTok; contKey = lnode; return true; }
    return false;
  }
  // _majorTrace with the recursion of _minorTrace unrolled onto a stack: an entry walks one 'worse' chain
  void majorTrace(Tok start, std::pair<unsigned, int> prevKey) {
    struct Fr { Tok tok; std::pair<unsigned, int> key; };
    std::vector<Fr> st; st.push_back(Fr{start, prevKey});
    while (!st.empty()) {
      Fr& fr = st.back();
      if (fr.tok < 0) { st.pop_back(); continue; }
      const Tok tok = fr.tok; const std::pair<unsigned, int> key = fr.key;
      // next of the chain: only placed tokens have a chain (epsilon tokens are never on the list)
      fr.tok = ((tok & 0xFFFF) == 0 && worse[(size_t) (tok >> 16)] >= 0) ? ((Tok) worse[(size_t) (tok >> 16)] << 16) : -1;
      Tok ct; std::pair<unsigned, int> ck;
      if (minorTrace(tok, key, ct, ck)) st.push_back(Fr{ct, ck});
    }
  }

  void run() {
    prepare();
    L.nodeFinal.assign(1, 0); edgesOf.assign(1, std::vector<int>());           // Lattice ctor: _initial = _newNode(0)
    L.finalStatesN = 0;
    if (I.haveNext) for (int i = 0; i < I.finN; i++) if (I.graph->nodes[I.fin[i].node].final_) L.finalStatesN++;
    auto tokOfBp = [&](int bp) -> Tok { return (Tok) I.arenaLat[bp] << 16; };
    if (L.finalStatesN > 0) {
      for (int i = 0; i < I.finN; i++) {
        const int node = I.fin[i].node;
        if (!I.graph->nodes[node].final_) continue;
        const std::pair<unsigned, int> init(stateIndex(node), I.T);            // (node->index(), _frameX + 1)
        const int ln = addNode(true); lnodes[init] = ln;
        majorTrace(tokOfBp(I.fin[i].bp), init);
      }
    } else if (I.finN > 0) {
      // _bestToken over the list (list order, strict '<' on the float score; decoder.h:639-685)
      int best = -1; double bestScore = HUGE_VAL;
      for (int i = 0; i < I.finN; i++) { const float s = f((uint32_t) I.fin[i].ac) + f((uint32_t) I.fin[i].lm); if ((double) s < bestScore) { bestScore = (double) s; best = i; } }
      if (best >= 0) {
        const std::pair<unsigned, int> init(stateIndex(I.fin[best].node), I.T);
        const int ln = addNode(false); lnodes[init] = ln;
        const int en = addNode(true);
        addEdge(ln, en, 0u, I.eosX, I.T, I.T, 0.0, 0.0);
        majorTrace(tokOfBp(I.fin[best].bp), init);
      }
    }
  }
};

}  // namespace

// _Decoder::writeGMM (decoder.h:1018-1102) up to the printing: runs of equal input symbols along the best token's chain (epsilon tokens
// skipped), in the order the reference collects them -- last run first.  The score column is what the shipped code computes: the acoustic
// score at the end of the run for the first row, the TOTAL score of the run's first token afterwards (:1074), minus the acoustic score of
// the token before the one the run stopped at.
bool best_path_gmm(const LatInput& in, std::vector<GmmRow>& rows)
{
  rows.clear();
  if (in.T <= 0 || in.finN <= 0) return false;
  LatticeData dummy; Builder b(in, dummy); b.prepare();
  int best = -1; double bestScore = HUGE_VAL;                                   // _bestToken (decoder.h:639-685): list order, strict '<'
  for (int i = 0; i < in.finN; i++) { const float s = Builder::f((uint32_t) in.fin[i].ac) + Builder::f((uint32_t) in.fin[i].lm); if ((double) s < bestScore) { bestScore = (double) s; best = i; } }
  if (best < 0) return false;
  Tok tok = (Tok) in.arenaLat[in.fin[best].bp] << 16;
  auto inOf = [&](Tok t) -> uint32_t { return in.csr->in[b.arcOf(t)]; };
  auto acOf = [&](Tok t) -> float { float a, l; b.scoresOf(t, a, l); return a; };
  auto scoreOf = [&](Tok t) -> float { float a, l; b.scoresOf(t, a, l); return a + l; };      // Token::score(): a float sum (lattice.h:57)
  Tok nextTok = tok; uint32_t thisX = inOf(tok); int endX = 0;
  while (tok >= 0 && thisX == 0) { nextTok = tok; tok = b.prevOf(tok); if (tok >= 0) thisX = inOf(tok); }
  double wscore = (double) acOf(nextTok);
  if (tok >= 0) endX = b.frameOfTok(tok);
  while (tok >= 0) {
    nextTok = tok; uint32_t inX = inOf(tok);
    while (tok >= 0 && inX == thisX) { nextTok = tok; tok = b.prevOf(tok); if (tok >= 0) inX = inOf(tok); }
    GmmRow r; r.inX = thisX; r.startX = b.frameOfTok(nextTok); r.endX = endX;
    Tok pp = tok >= 0 ? b.prevOf(tok) : (Tok) -1;
    const double oscore = (tok < 0 || pp < 0) ? 0.0 : (double) acOf(pp);
    r.score = wscore - oscore; rows.push_back(r);
    wscore = (double) scoreOf(nextTok);
    thisX = inX;
    while (tok >= 0 && thisX == 0) { tok = b.prevOf(tok); if (tok >= 0) thisX = inOf(tok); }
    if (tok >= 0) endX = b.frameOfTok(tok);
  }
  return true;
}

void build_lattice(const LatInput& in, LatticeData& out)
{
  out = LatticeData();
  if (in.T <= 0) { out.nodeFinal.assign(1, 0); return; }
  Builder b(in, out); b.run();
}

// flat image: [magic, nNodes, nEdges, finalStatesN] int32, nodeFinal[nNodes] int32, then per edge field arrays
std::vector<unsigned char> LatticeData::pack() const
{
  const int32_t hdr[4] = { 0x4C415431, (int32_t) nodeFinal.size(), (int32_t) from.size(), finalStatesN };
  const size_t nE = from.size(), nN = nodeFinal.size();
  std::vector<unsigned char> b(sizeof(hdr) + 4 * nN + nE * (6 * 4 + 2 * 8));
  unsigned char* q = b.data();
  auto put = [&](const void* p, size_t n) { if (n) memcpy(q, p, n); q += n; };
  put(hdr, sizeof(hdr)); put(nodeFinal.data(), 4 * nN); put(from.data(), 4 * nE); put(to.data(), 4 * nE); put(in.data(), 4 * nE); put(out.data(), 4 * nE);
  put(start.data(), 4 * nE); put(end.data(), 4 * nE); put(ac.data(), 8 * nE); put(lm.data(), 8 * nE);
  return b;
}
LatticeData LatticeData::unpack(const unsigned char* p, size_t n)
{
  LatticeData L; int32_t hdr[4];
  if (n < sizeof(hdr)) throw Error(DSR_E_PARSE, "lattice image too short");
  memcpy(hdr, p, sizeof(hdr));
  if (hdr[0] != 0x4C415431 || hdr[1] < 0 || hdr[2] < 0) throw Error(DSR_E_PARSE, "not a lattice image");
  const size_t nN = (size_t) hdr[1], nE = (size_t) hdr[2];
  if (n < sizeof(hdr) + 4 * nN + nE * 40) throw Error(DSR_E_PARSE, "lattice image truncated");
  const unsigned char* q = p + sizeof(hdr);
  auto get = [&](void* d, size_t k) { if (k) memcpy(d, q, k); q += k; };
  L.finalStatesN = hdr[3]; L.nodeFinal.resize(nN); L.from.resize(nE); L.to.resize(nE); L.in.resize(nE); L.out.resize(nE); L.start.resize(nE); L.end.resize(nE); L.ac.resize(nE); L.lm.resize(nE);
  get(L.nodeFinal.data(), 4 * nN); get(L.from.data(), 4 * nE); get(L.to.data(), 4 * nE); get(L.in.data(), 4 * nE); get(L.out.data(), 4 * nE);
  get(L.start.data(), 4 * nE); get(L.end.data(), 4 * nE); get(L.ac.data(), 8 * nE); get(L.lm.data(), 8 * nE);
  return L;
}

}  // namespace dsr
