/*
 * oracle/orc_wfst.c -- TEST INFRASTRUCTURE (see orc.h).
 * CPU restatement of the static decoding graph and the Viterbi token-passing decoder:
 *   asr/decoder/wfstFlyWeight.cc:63-118    _addFinal, find
 *   asr/decoder/wfstFlyWeight.cc:299-365   _readText (numeric AT&T form)
 *   asr/decoder/wfstFlyWeight.cc:367-413   _readBinary
 *   asr/decoder/wfstFlyWeight.cc:415-463   write
 *   asr/decoder/wfstFlyWeight.cc:552-556   Node::_addEdgeForce (prepend => reverse file order)
 *   asr/lattice/lattice.h:37-79            _Token (float ac/lm, float score())
 *   asr/decoder/decoder.h:206-247          _TokenList::insert/replace (LIFO list, replace in place)
 *   asr/decoder/decoder.h:488-545          _newUtterance, _expandToEnd, _placeOnList
 *   asr/decoder/decoder.h:548-595          _processFirstFrame, _processFrame
 *   asr/decoder/decoder.h:639-737          _bestToken, decode
 *   asr/decoder/decoder.h:748-773          bestHypo
 *   asr/decoder/decoder.h:956-1015         _expandNode, _expandNodeToEnd
 *   asr/decoder/decoder.h:517-545          _placeOnList incl. the lattice 'worse' chains (built always: decoder.h:1113-1114,1133-1135)
 *   asr/decoder/decoder.h:598-608,805-953  finalStatesN, lattice, _majorTrace, _minorTrace, _findLNode
 *   asr/lattice/lattice.cc:64-69,715-757,858-887  Lattice ctor (initial node 0), write, _topoSort/_visitNode
 *   asr/fsm/fsm.cc:122-138,153-178,541-545,1171-1178,553-559  _addFinal, _find, _addEdgeForce (prepend), Edge::write, Node::write
 *   asr/lattice/lattice.h:151-154          LatticeEdgeData::write
 * Token scores are stored as float and compared as the reference does (double candidate
 * against the incumbent's float sum).  Compile with -ffp-contract=off.
 */
#include "orc.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define END_MARKER 2147483647

typedef struct { int src, dst; unsigned in, out; float cost; int nextArc; } arc_t;
typedef struct { unsigned state; int final; float cost; int firstArc; int inFinalMap; int inNodeMap; } node_t;

struct orc_wfst {
  node_t* nodes; int nNodes, capNodes;
  arc_t* arcs; int nArcs, capArcs;
  int* nodeOf; unsigned nodeOfCap;    /* state -> node id for _nodes/_final maps, -1 if none */
  int initial;                         /* node id of _initial or -1 */
};

orc_wfst* orc_wfst_new(void)
{
  orc_wfst* g = (orc_wfst*) calloc(1, sizeof(orc_wfst));
  g->initial = -1; return g;
}
void orc_wfst_free(orc_wfst* g) { if (!g) return; free(g->nodes); free(g->arcs); free(g->nodeOf); free(g); }

static int new_node(orc_wfst* g, unsigned state)
{
  if (g->nNodes == g->capNodes) { g->capNodes = g->capNodes ? 2 * g->capNodes : 1024; g->nodes = (node_t*) realloc(g->nodes, sizeof(node_t) * g->capNodes); }
  node_t* n = &g->nodes[g->nNodes];
  n->state = state & 0x1FFFFFFFu;      /* 29-bit index field, wfstFlyWeight.h:181-183 */
  n->final = 0; n->cost = 0.0f; n->firstArc = -1; n->inFinalMap = 0; n->inNodeMap = 0;
  return g->nNodes++;
}
static void map_grow(orc_wfst* g, unsigned state)
{
  if (state < g->nodeOfCap) return;
  unsigned nc = g->nodeOfCap ? g->nodeOfCap : 1024; while (nc <= state) nc *= 2;
  g->nodeOf = (int*) realloc(g->nodeOf, sizeof(int) * nc);
  for (unsigned i = g->nodeOfCap; i < nc; i++) g->nodeOf[i] = -1;
  g->nodeOfCap = nc;
}
static int find_node(orc_wfst* g, unsigned state, int create)
{
  /* wfstFlyWeight.cc:94-118: initial first, then _nodes, then _final */
  if (g->initial >= 0 && g->nodes[g->initial].state == state) return g->initial;
  map_grow(g, state);
  if (g->nodeOf[state] >= 0) return g->nodeOf[state];
  if (!create) return -1;
  int id = new_node(g, state); g->nodes[id].inNodeMap = 1; g->nodeOf[state] = id; return id;
}
int orc_wfst_add_final(orc_wfst* g, unsigned state, float cost)
{
  /* wfstFlyWeight.cc:72-92 (does not consult _initial) */
  map_grow(g, state);
  int id = g->nodeOf[state];
  if (id >= 0 && g->nodes[id].inFinalMap) return -1;     /* "already has final node" */
  if (id < 0) { id = new_node(g, state); g->nodeOf[state] = id; }
  g->nodes[id].cost = cost; g->nodes[id].final = 1; g->nodes[id].inFinalMap = 1; g->nodes[id].inNodeMap = 0;
  return 0;
}
static int add_arc_raw(orc_wfst* g, unsigned s1, unsigned s2, unsigned in, unsigned out, float cost, int dropEpsSelf)
{
  int from;
  if (g->initial < 0) { from = new_node(g, s1); g->initial = from; }
  else from = find_node(g, s1, 1);
  int to = find_node(g, s2, 1);
  if (dropEpsSelf && s1 == s2 && in == 0 && out == 0) return 0;     /* :353 (text reader only) */
  if (g->nArcs == g->capArcs) { g->capArcs = g->capArcs ? 2 * g->capArcs : 4096; g->arcs = (arc_t*) realloc(g->arcs, sizeof(arc_t) * g->capArcs); }
  arc_t* a = &g->arcs[g->nArcs];
  a->src = from; a->dst = to; a->in = in; a->out = out; a->cost = cost;
  a->nextArc = g->nodes[from].firstArc; g->nodes[from].firstArc = g->nArcs;   /* prepend */
  g->nArcs++;
  return 0;
}
int orc_wfst_add_arc(orc_wfst* g, unsigned s1, unsigned s2, unsigned in, unsigned out, float cost)
{ return add_arc_raw(g, s1, s2, in, out, cost, 1); }

/* ---- big-endian primitives (btk/common/mach_ind_io.cc:176-200,331-350) ---- */
static int rd_int(FILE* fp, int* v) { unsigned char b[4]; if (fread(b, 1, 4, fp) != 4) return -1; *v = (int) ((unsigned) b[0] << 24 | (unsigned) b[1] << 16 | (unsigned) b[2] << 8 | (unsigned) b[3]); return 0; }
static int rd_float(FILE* fp, float* v) { int i; if (rd_int(fp, &i)) return -1; memcpy(v, &i, 4); return 0; }
static void wr_int(FILE* fp, int v) { unsigned u = (unsigned) v; unsigned char b[4] = { (unsigned char)(u >> 24), (unsigned char)(u >> 16), (unsigned char)(u >> 8), (unsigned char) u }; fwrite(b, 1, 4, fp); }
static void wr_float(FILE* fp, float v) { int i; memcpy(&i, &v, 4); wr_int(fp, i); }

static int read_impl(orc_wfst* g, const char* file, int binary, int noSelfLoops);
int orc_wfst_read(orc_wfst* g, const char* file, int binary) { return read_impl(g, file, binary, 0); }
/* WFSTransducer::read(fileName, noSelfLoops) (asr/fsm/fsm.cc:901-986) */
int orc_wfst_read_dynamic(orc_wfst* g, const char* file, int noSelfLoops) { return read_impl(g, file, 0, noSelfLoops); }
static int read_impl(orc_wfst* g, const char* file, int binary, int noSelfLoops)
{
  FILE* fp = fopen(file, binary ? "rb" : "r");
  if (!fp) return -7;   /* JIO */
  if (binary) {
    int n;
    while (rd_int(fp, &n) == 0 && n != END_MARKER) {
      if (n == 3) { int idx, end; float cost; rd_int(fp, &idx); rd_float(fp, &cost); rd_int(fp, &end); if (end != END_MARKER) { fclose(fp); return -7; } if (orc_wfst_add_final(g, (unsigned) idx, cost)) { fclose(fp); return -3; } }
      else if (n == 6) { int s1, s2, in, out, end; float cost; rd_int(fp, &s1); rd_int(fp, &s2); rd_int(fp, &in); rd_int(fp, &out); rd_float(fp, &cost); rd_int(fp, &end); if (end != END_MARKER) { fclose(fp); return -7; } add_arc_raw(g, (unsigned) s1, (unsigned) s2, (unsigned) in, (unsigned) out, cost, 0); }
      else { fclose(fp); return -7; }
    }
  } else {
    char* line = NULL; size_t cap = 0;
    while (getline(&line, &cap, fp) > 0) {
      char* tok[6]; int i = 0;
      tok[0] = strtok(line, " \t\n");
      if (!tok[0]) continue;   /* the reference would dereference NULL on a blank line */
      while ((i < 5) && ((tok[++i] = strtok(NULL, " \t\n")) != NULL));
      unsigned s1 = (unsigned) strtoul(tok[0], NULL, 0);
      if (i == 1) { if (orc_wfst_add_final(g, s1, 0.0f)) { free(line); fclose(fp); return -3; } }
      else if (i == 2) { float cost; sscanf(tok[1], "%f", &cost); if (orc_wfst_add_final(g, s1, cost)) { free(line); fclose(fp); return -3; } }
      else if (i == 4 || i == 5) {
        unsigned s2 = (unsigned) strtoul(tok[1], NULL, 0), in = (unsigned) strtoul(tok[2], NULL, 0), out = (unsigned) strtoul(tok[3], NULL, 0);
        if (s1 == s2 && noSelfLoops) continue;             /* fsm.cc:945 */
        float cost = 0.0f; if (i == 5) sscanf(tok[4], "%f", &cost);
        add_arc_raw(g, s1, s2, in, out, cost, 1);
      } else { free(line); fclose(fp); return -7; }
    }
    free(line);
  }
  fclose(fp);
  return 0;
}

static void write_arc(FILE* fp, const orc_wfst* g, const arc_t* a, int binary)
{
  if (binary) { wr_int(fp, 6); wr_int(fp, (int) g->nodes[a->src].state); wr_int(fp, (int) g->nodes[a->dst].state); wr_int(fp, (int) a->in); wr_int(fp, (int) a->out); wr_float(fp, a->cost); wr_int(fp, END_MARKER); }
  else {   /* wfstFlyWeight.cc:486-493 */
    fprintf(fp, "%10d  %10d  %10d  %10d", (int) g->nodes[a->src].state, (int) g->nodes[a->dst].state, (int) a->in, (int) a->out);
    if (a->cost == 0.0) fprintf(fp, "\n"); else fprintf(fp, "  %12g\n", (double) a->cost);
  }
}
static int cmp_state(const void* a, const void* b) { unsigned x = ((const unsigned*) a)[0], y = ((const unsigned*) b)[0]; return (x > y) - (x < y); }
int orc_wfst_write(const orc_wfst* g, const char* file, int binary)
{
  /* initial node's arcs, then _nodes in map (state) order, then _final in map order with node records */
  FILE* fp = fopen(file, binary ? "wb" : "w"); if (!fp) return -7;
  if (g->initial >= 0) for (int a = g->nodes[g->initial].firstArc; a >= 0; a = g->arcs[a].nextArc) write_arc(fp, g, &g->arcs[a], binary);
  unsigned* order = (unsigned*) malloc(sizeof(unsigned) * 2 * (size_t) (g->nNodes > 0 ? g->nNodes : 1)); int n = 0;
  for (int i = 0; i < g->nNodes; i++) if (g->nodes[i].inNodeMap) { order[2*n] = g->nodes[i].state; order[2*n+1] = (unsigned) i; n++; }
  qsort(order, n, 2 * sizeof(unsigned), cmp_state);
  for (int k = 0; k < n; k++) for (int a = g->nodes[order[2*k+1]].firstArc; a >= 0; a = g->arcs[a].nextArc) write_arc(fp, g, &g->arcs[a], binary);
  n = 0;
  for (int i = 0; i < g->nNodes; i++) if (g->nodes[i].inFinalMap) { order[2*n] = g->nodes[i].state; order[2*n+1] = (unsigned) i; n++; }
  qsort(order, n, 2 * sizeof(unsigned), cmp_state);
  for (int k = 0; k < n; k++) {
    const node_t* nd = &g->nodes[order[2*k+1]];
    for (int a = nd->firstArc; a >= 0; a = g->arcs[a].nextArc) write_arc(fp, g, &g->arcs[a], binary);
    if (binary) { wr_int(fp, 3); wr_int(fp, (int) nd->state); wr_float(fp, nd->cost); wr_int(fp, END_MARKER); }
    else { if (nd->cost == 0.0) fprintf(fp, "%10d\n", (int) nd->state); else fprintf(fp, "%10d  %12g\n", (int) nd->state, (double) nd->cost); }
  }
  if (binary) wr_int(fp, END_MARKER);
  free(order); fclose(fp);
  return 0;
}

int orc_wfst_num_nodes(const orc_wfst* g) { return g->nNodes; }
int orc_wfst_num_arcs(const orc_wfst* g) { return g->nArcs; }

/* CSR in iteration order.  csrOf[a] maps internal arc id -> exported arc id. */
static int* build_csr(const orc_wfst* g, int* arcOff)
{
  int* csrOf = (int*) malloc(sizeof(int) * (size_t) (g->nArcs > 0 ? g->nArcs : 1)); int pos = 0;
  for (int n = 0; n < g->nNodes; n++) { arcOff[n] = pos; for (int a = g->nodes[n].firstArc; a >= 0; a = g->arcs[a].nextArc) csrOf[a] = pos++; }
  arcOff[g->nNodes] = pos;
  return csrOf;
}
void orc_wfst_export(const orc_wfst* g, unsigned* nodeState, int* nodeFinal, float* nodeCost,
                     int* arcOff, int* arcDst, unsigned* arcIn, unsigned* arcOut, float* arcCost)
{
  int* csrOf = build_csr(g, arcOff);
  for (int n = 0; n < g->nNodes; n++) { nodeState[n] = g->nodes[n].state; nodeFinal[n] = g->nodes[n].final; nodeCost[n] = g->nodes[n].cost; }
  for (int a = 0; a < g->nArcs; a++) { int c = csrOf[a]; arcDst[c] = g->arcs[a].dst; arcIn[c] = g->arcs[a].in; arcOut[c] = g->arcs[a].out; arcCost[c] = g->arcs[a].cost; }
  free(csrOf);
}

/* ------------------------------ decoder ------------------------------ */
typedef struct { float ac, lm; int frame; int arc; int prev; int worse; } tok_t;       /* lattice.h:73-78 */
typedef struct { int tok; int state; int next; } holder_t;                    /* decoder.h:58-76 */
typedef struct {
  holder_t* h; int nH, capH; int head; int* ofState; int* stamp; int gen; int active;
} tlist_t;
typedef struct {
  const orc_wfst* g; const orc_dec_cfg* cfg; const float* scores; int T, nDist;
  tok_t* tok; int nTok, capTok;
  tlist_t *cur, *nxt; double topScore; int frameX; int ended; long activeHypos;
} dec_t;

static void tl_init(tlist_t* l, int nNodes) { memset(l, 0, sizeof(*l)); l->head = -1; l->ofState = (int*) malloc(sizeof(int) * (size_t) nNodes); l->stamp = (int*) calloc((size_t) nNodes, sizeof(int)); l->gen = 1; }
static void tl_free(tlist_t* l) { free(l->h); free(l->ofState); free(l->stamp); }
static void tl_clear(tlist_t* l) { l->nH = 0; l->head = -1; l->gen++; l->active = 0; }
static int tl_find(const tlist_t* l, int node) { return (l->stamp[node] == l->gen) ? l->ofState[node] : -1; }
static void tl_insert(tlist_t* l, int node, int tok)
{
  if (l->nH == l->capH) { l->capH = l->capH ? 2 * l->capH : 4096; l->h = (holder_t*) realloc(l->h, sizeof(holder_t) * l->capH); }
  holder_t* h = &l->h[l->nH]; h->tok = tok; h->state = node; h->next = l->head; l->head = l->nH;   /* LIFO, decoder.h:246-247 */
  l->ofState[node] = l->nH; l->stamp[node] = l->gen; l->nH++; l->active++;
}
static int new_tok(dec_t* d, double ac, double lm, int frame, int arc, int prev)
{
  if (d->nTok == d->capTok) { d->capTok = d->capTok ? 2 * d->capTok : (1 << 16); d->tok = (tok_t*) realloc(d->tok, sizeof(tok_t) * (size_t) d->capTok); }
  tok_t* t = &d->tok[d->nTok]; t->ac = (float) ac; t->lm = (float) lm; t->frame = frame; t->arc = arc; t->prev = prev; t->worse = -1;
  return d->nTok++;
}
static inline float tok_score(const tok_t* t) { return t->ac + t->lm; }       /* float sum, lattice.h:53 */

static void place_on_list(dec_t* d, int arc, double acScore, double lmScore, int thisToken)
{
  /* decoder.h:517-545.  The 'worse' chains of lattice generation: a better arrival takes the place on the list and hangs the
     incumbent (with its own chain) behind itself; a worse one is hung directly behind the incumbent, in front of its chain. */
  double ttlScore = acScore + lmScore;
  const arc_t* e = &d->g->arcs[arc];
  if (ttlScore < d->topScore && e->in != 0) d->topScore = ttlScore;
  int h = tl_find(d->nxt, e->dst);
  if (h >= 0) {
    const int inc = d->nxt->h[h].tok;
    if (ttlScore < tok_score(&d->tok[inc])) {
      const int nt = new_tok(d, acScore, lmScore, d->frameX, arc, thisToken);
      d->tok[nt].worse = inc; d->nxt->h[h].tok = nt;
    } else {
      const int wt = new_tok(d, acScore, lmScore, d->frameX, arc, thisToken);
      d->tok[wt].worse = d->tok[inc].worse; d->tok[inc].worse = wt;
    }
  } else tl_insert(d->nxt, e->dst, new_tok(d, acScore, lmScore, d->frameX, arc, thisToken));
}

static void expand_node(dec_t* d, int node, int thisToken)
{
  /* decoder.h:956-989 */
  double acScoreNode = 0.0, lmScoreNode = 0.0;
  if (thisToken >= 0) { acScoreNode = d->tok[thisToken].ac; lmScoreNode = d->tok[thisToken].lm; }
  for (int a = d->g->nodes[node].firstArc; a >= 0; a = d->g->arcs[a].nextArc) {
    if (d->ended) return;
    const arc_t* e = &d->g->arcs[a];
    unsigned distX = e->in;
    double lmScore = lmScoreNode + d->cfg->lmScale * (float) e->cost;
    if (e->out != 0) lmScore += (d->cfg->lmScale * d->cfg->lmPenalty);
    if (e->in == d->cfg->silenceX && (thisToken < 0 || d->g->arcs[d->tok[thisToken].arc].in != d->cfg->silenceX))
      lmScore += (d->cfg->lmScale * d->cfg->silPenalty);
    if (distX == 0) { expand_node(d, e->dst, new_tok(d, acScoreNode, lmScore, d->frameX, a, thisToken)); continue; }
    if (d->frameX >= d->T) { d->ended = 1; return; }     /* feature stream throws jiterator_error */
    double acScore = acScoreNode + d->scores[(size_t) d->frameX * d->nDist + (distX - 1)];
    place_on_list(d, a, acScore, lmScore, thisToken);
  }
}

static void expand_node_to_end(dec_t* d, int node, int thisToken)
{
  /* decoder.h:992-1015 */
  double acScoreNode = d->tok[thisToken].ac, lmScoreNode = d->tok[thisToken].lm;
  for (int a = d->g->nodes[node].firstArc; a >= 0; a = d->g->arcs[a].nextArc) {
    const arc_t* e = &d->g->arcs[a];
    if (e->in != 0) continue;
    double lmScore = lmScoreNode + d->cfg->lmScale * (float) e->cost;
    if (e->out != 0) lmScore += (d->cfg->lmScale * d->cfg->lmPenalty);
    if (e->in == d->cfg->silenceX && (d->g->arcs[d->tok[thisToken].arc].in != d->cfg->silenceX))
      lmScore += (d->cfg->lmScale * d->cfg->silPenalty);
    if (d->g->nodes[e->dst].final)
      place_on_list(d, a, acScoreNode, lmScore + d->cfg->lmScale * (float) d->g->nodes[node].cost, thisToken);
    expand_node_to_end(d, e->dst, new_tok(d, acScoreNode, lmScore, d->frameX, a, thisToken));
  }
}

static int best_token(const dec_t* d, int* reachedFinal)
{
  /* decoder.h:639-685: list order, strict '<' on the float score */
  int best = -1; double bestScore = HUGE_VAL;
  for (int h = d->nxt->head; h >= 0; h = d->nxt->h[h].next) {
    double s = tok_score(&d->tok[d->nxt->h[h].tok]);
    if (s < bestScore) { bestScore = s; best = d->nxt->h[h].tok; }
  }
  *reachedFinal = (best >= 0);
  if (best < 0) {
    bestScore = HUGE_VAL;
    for (int h = d->cur->head; h >= 0; h = d->cur->h[h].next) {
      double s = tok_score(&d->tok[d->cur->h[h].tok]);
      if (s < bestScore) { bestScore = s; best = d->cur->h[h].tok; }
    }
  }
  return best;
}

static void dump_list(orc_dec_result* res, const dec_t* d, const int* csrOf, int frame)
{
  long need = res->dumpN + d->nxt->active;
  if (need > res->dumpCap) {
    res->dumpCap = need * 2 + 1024;
    res->dumpNode = (int*) realloc(res->dumpNode, sizeof(int) * (size_t) res->dumpCap);
    res->dumpArc = (int*) realloc(res->dumpArc, sizeof(int) * (size_t) res->dumpCap);
    res->dumpAc = (float*) realloc(res->dumpAc, sizeof(float) * (size_t) res->dumpCap);
    res->dumpLm = (float*) realloc(res->dumpLm, sizeof(float) * (size_t) res->dumpCap);
  }
  res->dumpOff[frame] = res->dumpN;
  for (int h = d->nxt->head; h >= 0; h = d->nxt->h[h].next) {
    const tok_t* t = &d->tok[d->nxt->h[h].tok];
    res->dumpNode[res->dumpN] = d->nxt->h[h].state; res->dumpArc[res->dumpN] = csrOf[t->arc];
    res->dumpAc[res->dumpN] = t->ac; res->dumpLm[res->dumpN] = t->lm; res->dumpN++;
  }
  res->dumpOff[frame + 1] = res->dumpN;
}

/* ------------------------------ lattice (decoder.h:805-953) ------------------------------ */
typedef struct { long long key; int node; } lnode_t;            /* (state index, frame) -> lattice node index */
typedef struct {
  const dec_t* d; orc_lattice* L;
  lnode_t* map; int nMap, capMap;
  int lnStateIndices;
} lat_t;
static void lat_add_node(orc_lattice* L, int idx, int final)
{
  if (idx >= L->capNodes) { int nc = L->capNodes ? L->capNodes : 64; while (nc <= idx) nc *= 2; L->nodeFinal = (int*) realloc(L->nodeFinal, sizeof(int) * (size_t) nc); L->nodeFirstEdge = (int*) realloc(L->nodeFirstEdge, sizeof(int) * (size_t) nc);
    for (int i = L->capNodes; i < nc; i++) { L->nodeFinal[i] = -1; L->nodeFirstEdge[i] = -1; } L->capNodes = nc; }
  if (idx >= L->nNodes) L->nNodes = idx + 1;
  L->nodeFinal[idx] = final;
}
static int lat_map_find(lat_t* t, long long key) { for (int i = t->nMap - 1; i >= 0; i--) if (t->map[i].key == key) return t->map[i].node; return -1; }
static void lat_map_put(lat_t* t, long long key, int node)
{ if (t->nMap == t->capMap) { t->capMap = t->capMap ? 2 * t->capMap : 256; t->map = (lnode_t*) realloc(t->map, sizeof(lnode_t) * (size_t) t->capMap); } t->map[t->nMap].key = key; t->map[t->nMap].node = node; t->nMap++; }
static long long lkey(unsigned state, int frame) { return ((long long) state << 32) | (unsigned) frame; }
static void lat_add_edge(orc_lattice* L, int from, int to, unsigned in, unsigned out, int start, int end, double ac, double lm)
{
  if (L->nEdges == L->capEdges) {
    L->capEdges = L->capEdges ? 2 * L->capEdges : 256; size_t n = (size_t) L->capEdges;
    L->from = (int*) realloc(L->from, sizeof(int) * n); L->to = (int*) realloc(L->to, sizeof(int) * n); L->in = (unsigned*) realloc(L->in, sizeof(unsigned) * n);
    L->out = (unsigned*) realloc(L->out, sizeof(unsigned) * n); L->start = (int*) realloc(L->start, sizeof(int) * n); L->end = (int*) realloc(L->end, sizeof(int) * n);
    L->ac = (double*) realloc(L->ac, sizeof(double) * n); L->lm = (double*) realloc(L->lm, sizeof(double) * n); L->nextEdge = (int*) realloc(L->nextEdge, sizeof(int) * n);
  }
  const int e = L->nEdges++;
  L->from[e] = from; L->to[e] = to; L->in[e] = in; L->out[e] = out; L->start[e] = start; L->end[e] = end; L->ac[e] = ac; L->lm[e] = lm;
  L->nextEdge[e] = L->nodeFirstEdge[from]; L->nodeFirstEdge[from] = e;          /* addEdgeForce: prepend (fsm.cc:541-545) */
}
static void major_trace(lat_t* t, int start, long long prev);
static void minor_trace(lat_t* t, int endTok, long long prev)
{
  /* decoder.h:873-931 */
  const dec_t* d = t->d; const orc_wfst* g = d->g;
  int tok = endTok; const int endFrame = d->tok[endTok].frame;
  unsigned currentInput = g->arcs[d->tok[tok].arc].in, currentOutput = g->arcs[d->tok[tok].arc].out;
  while (d->tok[tok].prev >= 0) {
    const arc_t* pe = &g->arcs[d->tok[d->tok[tok].prev].arc];
    if (pe->out != 0 && currentOutput != 0) break;
    if (pe->in != 0) { if (currentInput != 0 && pe->in != currentInput) break; currentInput = pe->in; }
    if (pe->out != 0) { if (currentOutput != 0) break; currentOutput = pe->out; }
    tok = d->tok[tok].prev;
  }
  const int begFrame = d->tok[tok].frame; const int prevTok = d->tok[tok].prev;
  const int endNode = lat_map_find(t, prev);                   /* _findLNode(lattice, prev): exists by construction */
  long long lnode = 0; int begNode; int create = 1;
  double acScore = d->tok[endTok].ac, lmScore = d->tok[endTok].lm;
  if (prevTok < 0) begNode = 0;                               /* lattice->initial() */
  else {
    acScore -= d->tok[prevTok].ac; lmScore -= d->tok[prevTok].lm;
    lnode = lkey(g->nodes[g->arcs[d->tok[tok].arc].src].state, begFrame);
    begNode = lat_map_find(t, lnode);
    if (begNode >= 0) create = 0;
    else { begNode = ++t->lnStateIndices; lat_add_node(t->L, begNode, 0); lat_map_put(t, lnode, begNode); }
  }
  if (currentInput == d->cfg->silenceX) lmScore -= (d->cfg->lmScale * d->cfg->silPenalty);
  if (currentOutput != 0) lmScore -= (d->cfg->lmScale * d->cfg->lmPenalty);
  lmScore /= d->cfg->lmScale;
  lat_add_edge(t->L, begNode, endNode, currentInput, currentOutput, begFrame, endFrame, acScore, lmScore);
  if (prevTok >= 0 && create) major_trace(t, prevTok, lnode);
}
static void major_trace(lat_t* t, int start, long long prev)
{ for (int tok = start; tok >= 0; tok = t->d->tok[tok].worse) minor_trace(t, tok, prev); }

static void build_lattice(const dec_t* d, orc_lattice* L, unsigned eosX)
{
  lat_t t; memset(&t, 0, sizeof(t)); t.d = d; t.L = L;
  const orc_wfst* g = d->g;
  lat_add_node(L, 0, 0);                                       /* Lattice ctor: _initial = _newNode(0) (lattice.cc:64-69) */
  int nFinal = 0;
  for (int h = d->nxt->head; h >= 0; h = d->nxt->h[h].next) if (g->nodes[g->arcs[d->tok[d->nxt->h[h].tok].arc].dst].final) nFinal++;
  if (nFinal > 0) {
    for (int h = d->nxt->head; h >= 0; h = d->nxt->h[h].next) {
      const int tk = d->nxt->h[h].tok; const int nd = g->arcs[d->tok[tk].arc].dst;
      if (!g->nodes[nd].final) continue;
      const long long init = lkey(g->nodes[nd].state, d->frameX + 1);
      const int ln = ++t.lnStateIndices; lat_add_node(L, ln, 1); lat_map_put(&t, init, ln);
      major_trace(&t, tk, init);
    }
  } else {
    int reached = 0; const int best = best_token(d, &reached);
    if (best >= 0) {
      const long long init = lkey(g->nodes[g->arcs[d->tok[best].arc].dst].state, d->frameX + 1);
      const int ln = ++t.lnStateIndices; lat_add_node(L, ln, 0); lat_map_put(&t, init, ln);
      const int en = ++t.lnStateIndices; lat_add_node(L, en, 1);
      lat_add_edge(L, ln, en, 0, eosX, d->frameX + 1, d->frameX + 1, 0.0, 0.0);
      major_trace(&t, best, init);
    }
  }
  free(t.map);
}
/* _Decoder::writeGMM (decoder.h:1018-1102) up to the printing: the runs of equal input symbols along the best token's chain, in the order the
   reference pushes them (last run first).  Row i: input symbol, first and last frame of the run, score as shipped -- the acoustic score at
   the end of the run (for the first row pushed; afterwards the TOTAL score of the run's first token: decoder.h:1074) minus the acoustic score
   of the token two before the run. */
static void build_gmm_rows(const dec_t* d, orc_gmm_rows* R)
{
  const orc_wfst* g = d->g;
  int reached = 0; int tok = best_token(d, &reached);
  memset(R, 0, sizeof(*R));
  if (tok < 0) { R->n = -1; return; }                          /* the reference dereferences a null token here */
  const int cap = 4096; int capN = cap;
  R->inX = (unsigned*) malloc(sizeof(unsigned) * capN); R->startX = (int*) malloc(sizeof(int) * capN); R->endX = (int*) malloc(sizeof(int) * capN);
  R->score = (double*) malloc(sizeof(double) * capN);
#define TIN(t) (g->arcs[d->tok[t].arc].in)
  int nextTok = tok; unsigned thisX = TIN(tok); int endX = 0; double wscore;
  while (tok >= 0 && thisX == 0) { nextTok = tok; tok = d->tok[tok].prev; if (tok >= 0) thisX = TIN(tok); }
  wscore = d->tok[nextTok].ac;
  if (tok >= 0) endX = d->tok[tok].frame;
  while (tok >= 0) {
    nextTok = tok; unsigned inX = TIN(tok);
    while (tok >= 0 && inX == thisX) { nextTok = tok; tok = d->tok[tok].prev; if (tok >= 0) inX = TIN(tok); }
    const int startX = d->tok[nextTok].frame;
    if (R->n == capN) { capN *= 2; R->inX = (unsigned*) realloc(R->inX, sizeof(unsigned) * capN); R->startX = (int*) realloc(R->startX, sizeof(int) * capN);
                        R->endX = (int*) realloc(R->endX, sizeof(int) * capN); R->score = (double*) realloc(R->score, sizeof(double) * capN); }
    const double oscore = (tok < 0 || d->tok[tok].prev < 0) ? 0.0 : d->tok[d->tok[tok].prev].ac;
    R->inX[R->n] = thisX; R->startX[R->n] = startX; R->endX[R->n] = endX; R->score[R->n] = wscore - oscore; R->n++;
    wscore = (float) (d->tok[nextTok].ac + d->tok[nextTok].lm);                /* Token::score() is a float sum (lattice.h:57) */
    thisX = inX;
    while (tok >= 0 && thisX == 0) { tok = d->tok[tok].prev; if (tok >= 0) thisX = TIN(tok); }
    if (tok >= 0) endX = d->tok[tok].frame;
  }
#undef TIN
}
void orc_gmm_rows_free(orc_gmm_rows* R) { if (!R) return; free(R->inX); free(R->startX); free(R->endX); free(R->score); memset(R, 0, sizeof(*R)); }

void orc_lattice_free(orc_lattice* L)
{ if (!L) return; free(L->nodeFinal); free(L->nodeFirstEdge); free(L->from); free(L->to); free(L->in); free(L->out); free(L->start); free(L->end); free(L->ac); free(L->lm); free(L->nextEdge); memset(L, 0, sizeof(*L)); }

/* Lattice::write(fileName, useSymbols = false, writeData) (lattice.cc:715-757): depth-first topological order from the initial node
   (finished nodes go to the front of the list), edges of the non-final nodes, then per final node (index order) its edges and its node line */
static void lat_visit(const orc_lattice* L, int node, int* color, int* order, int* nOrder, int* err)
{
  if (color[node] == 2) return;
  if (color[node] == 1) { *err = 1; return; }                  /* "graph is not acyclic" */
  color[node] = 1;
  for (int e = L->nodeFirstEdge[node]; e >= 0 && !*err; e = L->nextEdge[e]) lat_visit(L, L->to[e], color, order, nOrder, err);
  color[node] = 2; order[(*nOrder)++] = node;                  /* reversed below = push_front */
}
static void lat_write_edge(FILE* fp, const orc_lattice* L, int e, int writeData)
{
  fprintf(fp, "%10d  %10d  %10d  %10d", L->from[e], L->to[e], (int) L->in[e], (int) L->out[e]);
  fprintf(fp, "\n");                                           /* edge cost is ZeroWeight */
  if (writeData) fprintf(fp, "%4d  %4d  %8.4f  %8.4f  %8.4f\n", L->start[e], L->end[e], L->ac[e], L->lm[e], 0.0);
}
int orc_lattice_write(const orc_lattice* L, const char* file, int writeData)
{
  FILE* fp = fopen(file, "w"); if (!fp) return -7;
  int* color = (int*) calloc((size_t) L->nNodes + 1, sizeof(int)); int* order = (int*) malloc(sizeof(int) * ((size_t) L->nNodes + 1)); int nOrder = 0, err = 0;
  lat_visit(L, 0, color, order, &nOrder, &err);
  if (!err) {
    for (int i = nOrder - 1; i >= 0; i--) { const int nd = order[i]; if (L->nodeFinal[nd] == 1) continue; for (int e = L->nodeFirstEdge[nd]; e >= 0; e = L->nextEdge[e]) lat_write_edge(fp, L, e, writeData); }
    for (int nd = 0; nd < L->nNodes; nd++) if (L->nodeFinal[nd] == 1) { for (int e = L->nodeFirstEdge[nd]; e >= 0; e = L->nextEdge[e]) lat_write_edge(fp, L, e, writeData); fprintf(fp, "%10d\n", nd); }
  }
  free(color); free(order); fclose(fp);
  return err ? -3 : 0;
}

int orc_decode_lat(const orc_wfst* g, const orc_dec_cfg* cfg, const float* scores, int T, int nDist, orc_dec_result* res, orc_lattice* lat, unsigned eosX);
int orc_decode_ex(const orc_wfst* g, const orc_dec_cfg* cfg, const float* scores, int T, int nDist, orc_dec_result* res, orc_lattice* lat, unsigned eosX, orc_gmm_rows* gmm);
int orc_decode(const orc_wfst* g, const orc_dec_cfg* cfg, const float* scores, int T, int nDist, orc_dec_result* res)
{ return orc_decode_lat(g, cfg, scores, T, nDist, res, NULL, 0); }
int orc_decode_lat(const orc_wfst* g, const orc_dec_cfg* cfg, const float* scores, int T, int nDist, orc_dec_result* res, orc_lattice* lat, unsigned eosX)
{ return orc_decode_ex(g, cfg, scores, T, nDist, res, lat, eosX, NULL); }
int orc_decode_ex(const orc_wfst* g, const orc_dec_cfg* cfg, const float* scores, int T, int nDist, orc_dec_result* res, orc_lattice* lat, unsigned eosX, orc_gmm_rows* gmm)
{
  memset(res, 0, sizeof(*res));
  if (g->initial < 0) return -3;
  int* arcOff = (int*) malloc(sizeof(int) * (size_t) (g->nNodes + 1));
  int* csrOf = build_csr(g, arcOff);
  dec_t d; memset(&d, 0, sizeof(d));
  tlist_t A, B; tl_init(&A, g->nNodes); tl_init(&B, g->nNodes);
  d.g = g; d.cfg = cfg; d.scores = scores; d.T = T; d.nDist = nDist; d.cur = &A; d.nxt = &B;
  res->activeCount = (int*) calloc((size_t) (T + 2), sizeof(int));
  res->topScore = (double*) calloc((size_t) (T + 2), sizeof(double));
  res->dumpOff = (long*) calloc((size_t) (T + 3), sizeof(long));

  /* _processFirstFrame */
  d.topScore = HUGE_VAL; d.frameX = 0;
  expand_node(&d, g->initial, -1);
  d.activeHypos = d.nxt->active;
  if (d.ended) {   /* no frames at all: the exception escapes decode() (decoder.h:691) */
    tl_free(&A); tl_free(&B); free(d.tok); free(arcOff); free(csrOf); return -8;
  }
  res->activeCount[0] = d.nxt->active; res->topScore[0] = d.topScore; res->nActive = 1;
  if (cfg->dumpTokens) dump_list(res, &d, csrOf, 0);
  /* frames */
  while (!d.ended) {
    d.frameX++;
    tlist_t* tmp = d.cur; d.cur = d.nxt; d.nxt = tmp; tl_clear(d.nxt);     /* _processFrame :564 */
    double thresh = d.topScore + cfg->beam;
    d.topScore = HUGE_VAL;
    int any = 0;
    if (cfg->topN > 0) {
      /* SortedIterator (decoder.h:298-320): holders in list order, sorted by the float score (insertion sort = stable: ties keep list order,
         which the reference's std::sort leaves open), the first topN are expanded, no beam (:571-581) */
      int cnt = 0; for (int h = d.cur->head; h >= 0; h = d.cur->h[h].next) cnt++;
      int* ord = (int*) malloc(sizeof(int) * (size_t) (cnt > 0 ? cnt : 1)); int k = 0;
      for (int h = d.cur->head; h >= 0; h = d.cur->h[h].next) {
        int tk = d.cur->h[h].tok; float sc_ = tok_score(&d.tok[tk]); int j = k++;
        while (j > 0 && tok_score(&d.tok[ord[j - 1]]) > sc_) { ord[j] = ord[j - 1]; j--; }
        ord[j] = tk;
      }
      for (int i = 0; i < cnt && i < cfg->topN && !d.ended; i++) { any = 1; expand_node(&d, g->arcs[d.tok[ord[i]].arc].dst, ord[i]); }
      free(ord);
    } else
    for (int h = d.cur->head; h >= 0 && !d.ended; h = d.cur->h[h].next) {
      int tk = d.cur->h[h].tok;
      double score = tok_score(&d.tok[tk]);
      if (score > thresh) continue;
      any = 1;
      expand_node(&d, g->arcs[d.tok[tk].arc].dst, tk);
    }
    if (d.ended) break;
    d.activeHypos += d.nxt->active;
    if (d.frameX >= T) {   /* nothing asked for a score at frame T: the reference never terminates */
      (void) any; tl_free(&A); tl_free(&B); free(d.tok); free(arcOff); free(csrOf); return -100;
    }
    res->activeCount[d.frameX] = d.nxt->active; res->topScore[d.frameX] = d.topScore; res->nActive = d.frameX + 1;
    if (cfg->dumpTokens) dump_list(res, &d, csrOf, d.frameX);
  }
  d.frameX--;
  /* _expandToEnd, decoder.h:500-513 */
  tl_clear(d.nxt);
  for (int h = d.cur->head; h >= 0; h = d.cur->h[h].next) {
    int tk = d.cur->h[h].tok; const tok_t* t = &d.tok[tk];
    int nd = g->arcs[t->arc].dst;
    if (g->nodes[nd].final) {
      float lmScore = t->lm + cfg->lmScale * (float) g->nodes[nd].cost;
      place_on_list(&d, t->arc, t->ac, lmScore, t->prev);
    }
    expand_node_to_end(&d, nd, tk);
  }
  int reached = 0; int best = best_token(&d, &reached);
  res->frames = d.frameX; res->reachedFinal = reached; res->activeHypos = d.activeHypos;
  if (best >= 0) {
    res->ac = d.tok[best].ac; res->lm = d.tok[best].lm; res->score = (double) res->ac + (double) res->lm;
    int n = 0; for (int t = best; t >= 0; t = d.tok[t].prev) n++;
    res->nArcs = n; res->arcs = (int*) malloc(sizeof(int) * (size_t) n); res->arcFrames = (int*) malloc(sizeof(int) * (size_t) n);
    res->words = (unsigned*) malloc(sizeof(unsigned) * (size_t) n);
    int i = n; for (int t = best; t >= 0; t = d.tok[t].prev) { i--; res->arcs[i] = csrOf[d.tok[t].arc]; res->arcFrames[i] = d.tok[t].frame; }
    /* bestHypo (decoder.h:748-773): outputs != 0 along the chain, first..last */
    { int k = n; unsigned* tmpw = (unsigned*) malloc(sizeof(unsigned) * (size_t) n);
      for (int t = best; t >= 0; t = d.tok[t].prev) { k--; tmpw[k] = g->arcs[d.tok[t].arc].out; }
      for (k = 0; k < n; k++) if (tmpw[k] != 0) res->words[res->nWords++] = tmpw[k];
      free(tmpw); }
  }
  res->finalStatesN = 0;
  for (int h = d.nxt->head; h >= 0; h = d.nxt->h[h].next) if (g->nodes[g->arcs[d.tok[d.nxt->h[h].tok].arc].dst].final) res->finalStatesN++;
  if (lat) { memset(lat, 0, sizeof(*lat)); build_lattice(&d, lat, eosX); }
  if (gmm) build_gmm_rows(&d, gmm);
  tl_free(&A); tl_free(&B); free(d.tok); free(arcOff); free(csrOf);
  return 0;
}

void orc_dec_result_free(orc_dec_result* r)
{ free(r->arcs); free(r->arcFrames); free(r->words); free(r->activeCount); free(r->topScore);
  free(r->dumpOff); free(r->dumpNode); free(r->dumpAc); free(r->dumpLm); free(r->dumpArc); memset(r, 0, sizeof(*r)); }
