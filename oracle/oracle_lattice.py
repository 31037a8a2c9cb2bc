"""oracle/oracle_lattice.py -- TEST INFRASTRUCTURE: CPU restatement of asr/lattice's Lattice operations.

Only tests/ may import this.  Pure-Python loops on purpose: the lattices of the tests hold hundreds to a few thousand links.

Follows the reference object by object (nodes with a linked edge list, the initial node / the vector _nodes / the map _final,
the cached list of sorted nodes, tokens with float scores), so that it shares no structure with the array-based product code:
  asr/lattice/lattice.cc   rescore :122-171, bestHypo :281-306, gammaProbs :309-379, writeCTM :420-477, writePhoneCTM :479-537,
                           writeHypoHTK :539-601, writeWordConfs :603-646, prune :648-693, pruneEdges :695-713, write :715-757,
                           purge :776-841, _topoSort :858-887, Node::_removeLinks :949-965, EdgeIterator :984-1002
  asr/lattice/lattice.h    _Token :38-78 (float scores), LatticeEdgeData :127-164
  asr/fsm/fsm.cc           logAdd :38-56, _addFinal :122-138, _find :153-178, setColor :428-441, Node::write :553-559,
                           Edge::write :1171-1178;  asr/fsm/fsm.h WFST::read :3787-3873
PARITY UNPINNED: the reference holds no lattice file, golden output or driver for these calls; what pins this module is the
reading of the code above and closed-form checks in tests/test_lattice_ops_cpu.py.
"""
import math
import re
import sys

import numpy as np

LogZero = 1.0E10
HUGE = float(np.float32(3.40282347e+38))
f32 = np.float32


def logAdd(ap, bp):
    if ap > LogZero:
        raise ValueError("ap (%g) > LogZero (%g)" % (ap, LogZero))
    if bp > LogZero:
        raise ValueError("bp (%g) > LogZero (%g)" % (bp, LogZero))
    if ap > bp:
        ap, bp = bp, ap
    diff = ap - bp
    try:
        z = math.exp(diff)
    except OverflowError:
        z = float("inf")
    if z != z:
        raise ValueError("ap - bp returned NaN.")
    return ap - math.log(1.0 + z)


class Token:
    __slots__ = ("ac", "lm", "edge", "prev")

    def __init__(self, acs, lms, edge, prev):
        self.ac = f32(acs); self.lm = f32(lms); self.edge = edge; self.prev = prev

    def score(self):
        return f32(self.ac + self.lm)


class Edge:
    __slots__ = ("prev", "next", "input", "output", "cost", "start", "end", "ac", "lm", "gamma", "link", "serial")

    def __init__(self, prev, nxt, inp, out, start=-1, end=-1, ac=0.0, lm=0.0, gamma=0.0, cost=0.0, serial=-1):
        self.prev = prev; self.next = nxt; self.input = inp; self.output = out; self.cost = f32(cost)
        self.start = start; self.end = end; self.ac = ac; self.lm = lm; self.gamma = gamma; self.link = None; self.serial = serial


class Node:
    __slots__ = ("index", "final", "cost", "color", "success", "edges", "fwd", "bwd", "tok", "serial")

    def __init__(self, idx, serial):
        self.index = idx; self.final = False; self.cost = f32(0.0); self.color = 0; self.success = False
        self.edges = None; self.fwd = LogZero; self.bwd = LogZero; self.tok = None; self.serial = serial

    def addEdgeForce(self, ed):
        ed.link = self.edges; self.edges = ed

    def iter(self):
        e = self.edges
        while e is not None:
            yield e
            e = e.link

    def removeLinks(self, threshold):
        prev = None; curr = self.edges
        while curr is not None:
            if curr.gamma > threshold:
                if prev is None:
                    self.edges = curr.link
                else:
                    prev.link = curr.link
            else:
                prev = curr
            curr = curr.link


class Lattice:
    def __init__(self):
        self.initial = None; self.nodes = []; self.final = {}; self.sortedNodes = []
        self.acScale = 1.0; self.lmScale = 1.0; self.lmPenalty = 0.0; self.silPenalty = 0.0; self.silenceX = 0
        self.latticeForwardProb = 0.0
        self.allMade = []          # every node ever made, creation order (serial) -- for reporting only
        self.allEdges = []

    # ------------------------------------------------------------------ construction
    def _newNode(self, state):
        n = Node(state, len(self.allMade)); self.allMade.append(n); return n

    def _find(self, state):
        if self.initial is None:
            self.initial = self._newNode(state); return self.initial
        if self.initial.index == state:
            return self.initial
        if state in self.final:
            return self.final[state]
        if state < len(self.nodes) and self.nodes[state] is not None:
            return self.nodes[state]
        while state >= len(self.nodes):
            self.nodes.append(None)
        self.nodes[state] = self._newNode(state)
        return self.nodes[state]

    def _addFinal(self, state, cost=0.0):
        if state in self.final:
            raise ValueError("Automaton already has final node %d." % state)
        if state >= len(self.nodes) or self.nodes[state] is None:
            nd = self._newNode(state)
        else:
            nd = self.nodes[state]; self.nodes[state] = None
        nd.cost = f32(cost); nd.final = True
        self.final[state] = nd

    def _finis(self):
        return [self.final[k] for k in sorted(self.final)]

    @staticmethod
    def from_arrays(d):
        """the lattice as the decoder (or the oracle's decoder) hands it over: node k prints as k; edges in creation order"""
        L = Lattice()
        n = len(d["nodeFinal"])
        L.initial = L._newNode(0)
        made = {0: L.initial}
        for k in range(1, n):
            if d["nodeFinal"][k] == 1:
                L._addFinal(k); made[k] = L.final[k]
            else:
                made[k] = L._find(k)
        for e in range(len(d["from"])):
            ed = Edge(made[int(d["from"][e])], made[int(d["to"][e])], int(d["in"][e]), int(d["out"][e]), int(d["start"][e]), int(d["end"][e]),
                      float(d["ac"][e]), float(d["lm"][e]), serial=e)
            ed.prev.addEdgeForce(ed); L.allEdges.append(ed)
        return L

    @staticmethod
    def read(fileName, noSelfLoops=False, readData=False, inlex=None, outlex=None):
        L = Lattice()
        buf = open(fileName, "r").read()
        pos = 0
        def sym(tok, lex):                                       # strtoul(token, &p, 0), else the lexicon
            m = re.match(r"\s*([+-]?)(0[xX][0-9a-fA-F]+|0[0-7]*|[1-9][0-9]*)", tok)
            if not m:
                return lex.index(tok)
            d = m.group(2)
            v = int(d, 16) if d[:2] in ("0x", "0X") else (int(d, 8) if d[0] == "0" and len(d) > 1 else int(d))
            return (-v) & 0xFFFFFFFF if m.group(1) == "-" else v
        while pos < len(buf):
            eol = buf.find("\n", pos)
            if eol < 0:
                eol = len(buf)
            line = buf[pos:eol]; pos = min(eol + 1, len(buf))
            tok = line.split()
            i = min(len(tok), 5)
            s1 = int(re.match(r"\d+", tok[0]).group(0))
            if i == 1 or i == 2:
                L._addFinal(s1, float(tok[1]) if i == 2 else 0.0)
            elif i == 4 or i == 5:
                s2 = int(re.match(r"\d+", tok[1]).group(0))
                if s1 == s2 and noSelfLoops:
                    continue
                frm = L._find(s1); to = L._find(s2)
                inp = sym(tok[2], inlex); out = sym(tok[3], outlex)
                if s1 == s2 and inp == 0:
                    continue
                cost = float(tok[4]) if i == 5 else 0.0
                ed = Edge(frm, to, inp, out, cost=cost, serial=len(L.allEdges))
                frm.addEdgeForce(ed); L.allEdges.append(ed)
                if readData:
                    m = re.match(r"\s*(-?\d+)\s+(-?\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s*", buf[pos:])
                    if not m:
                        raise IOError("Only matched fewer than 5 elements.")
                    ed.start = int(m.group(1)); ed.end = int(m.group(2)); ed.ac = float(m.group(3)); ed.lm = float(m.group(4)); ed.gamma = float(m.group(5))
                    pos += m.end()
            else:
                raise IOError("Transducer file %s is inconsistent." % fileName)
        return L

    # ------------------------------------------------------------------ topological order
    def _setColor(self, c):
        if self.initial is not None:
            self.initial.color = c
        for nd in self.nodes:
            if nd is not None:
                nd.color = c
        for nd in self.final.values():
            nd.color = c

    def _topoSort(self):
        if len(self.sortedNodes) > 0:
            return
        self._setColor(0)
        # _visitNode, recursion unrolled: (node, iterator); a finished node goes to the FRONT of the list
        order = []
        def enter(nd, stack):
            if nd.color == 2:
                return
            if nd.color == 1:
                raise ValueError("Node %d is gray; graph is not acyclic." % nd.index)
            nd.color = 1; stack.append((nd, nd.iter()))
        stack = []; enter(self.initial, stack)
        while stack:
            nd, it = stack[-1]
            nxt = next(it, None)
            if nxt is None:
                nd.color = 2; order.append(nd); stack.pop()
            else:
                enter(nxt.next, stack)
        self.sortedNodes = order[::-1]

    def _clearSorted(self):
        self.sortedNodes = []

    # ------------------------------------------------------------------ rescoring
    def _link_lm(self, node, edge):
        lmScore = self.lmScale * edge.lm
        if edge.output != 0:
            lmScore += self.lmScale * self.lmPenalty
        if edge.input == self.silenceX and (node.tok is None or node.tok.edge.input != self.silenceX):
            lmScore += self.lmScale * self.silPenalty
        return lmScore

    def rescore(self, lmScale=30.0, lmPenalty=0.0, silPenalty=0.0, silenceX=0):
        self.lmScale = lmScale; self.lmPenalty = lmPenalty; self.silPenalty = silPenalty; self.silenceX = silenceX
        self._topoSort()
        for nd in self.sortedNodes:
            nd.tok = None
        for node in self.sortedNodes:
            acScoreNode = 0.0; lmScoreNode = 0.0
            if node.tok is not None:
                acScoreNode = float(node.tok.ac); lmScoreNode = float(node.tok.lm)
            for edge in node.iter():
                acScore = acScoreNode + self.acScale * edge.ac
                lmScore = lmScoreNode + self.lmScale * edge.lm
                if edge.output != 0:
                    lmScore += self.lmScale * self.lmPenalty
                if edge.input == self.silenceX and (node.tok is None or node.tok.edge.input != self.silenceX):
                    lmScore += self.lmScale * self.silPenalty
                ttlScore = acScore + lmScore
                if edge.next.tok is None or ttlScore < float(edge.next.tok.score()):
                    edge.next.tok = Token(acScore, lmScore, edge, node.tok)
        return self._bestToken().score()

    def _bestToken(self):
        bestTok = None; bestScore = HUGE
        for nd in self._finis():
            if nd.tok is not None and float(nd.tok.score()) < bestScore:
                bestTok = nd.tok; bestScore = float(nd.tok.score())
        return bestTok

    def bestHypo(self, useInputSymbols=False):
        tok = self._bestToken(); lastX = 0; hypo = []
        while True:
            if useInputSymbols:
                inX = tok.edge.input
                if inX != 0 and inX != lastX:
                    hypo.insert(0, inX); lastX = inX
            else:
                if tok.edge.output != 0:
                    hypo.insert(0, tok.edge.output)
            tok = tok.prev
            if tok is None:
                break
        return hypo

    # ------------------------------------------------------------------ posteriors
    def gammaProbs(self, acScale=1.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silenceX=0):
        self.acScale = acScale; self.lmScale = lmScale; self.lmPenalty = lmPenalty; self.silPenalty = silPenalty; self.silenceX = silenceX
        self._topoSort()
        self.initial.fwd = 0.0; self.initial.bwd = LogZero
        for nd in self.nodes:
            if nd is not None:
                nd.fwd = LogZero; nd.bwd = LogZero
        for nd in self.final.values():
            nd.fwd = LogZero; nd.bwd = 0.0
        for node in self.sortedNodes:                                   # _forwardProbs
            scoreNode = node.fwd
            for edge in node.iter():
                acScore = self.acScale * edge.ac
                lmScore = self._link_lm(node, edge)
                edge.next.fwd = logAdd(scoreNode + acScore + lmScore, edge.next.fwd)
        self.latticeForwardProb = LogZero
        for nd in self._finis():
            self.latticeForwardProb = logAdd(self.latticeForwardProb, nd.fwd)
        for node in reversed(self.sortedNodes):                         # _backwardProbs
            for edge in node.iter():
                acScore = self.acScale * edge.ac
                lmScore = self._link_lm(node, edge)
                ttlScore = edge.next.bwd + acScore + lmScore
                if ttlScore >= LogZero:
                    continue
                node.bwd = logAdd(ttlScore, node.bwd)
        back = self.initial.bwd
        if (abs(back - self.latticeForwardProb) / self.latticeForwardProb) > 0.0001:
            raise ValueError("Lattice forward (%g) and backward probabilities (%g) are not equal." % (self.latticeForwardProb, back))
        for node in self.sortedNodes:                                   # _gammaProbs
            for edge in node.iter():
                acScore = self.acScale * edge.ac
                lmScore = self._link_lm(node, edge)
                gamma = node.fwd + acScore + lmScore + edge.next.bwd - self.latticeForwardProb
                if gamma < 0.0:
                    if gamma < -0.0001:
                        raise ValueError("Neg. Log-Probability (%g)" % gamma)
                    gamma = 0.0
                edge.gamma = gamma
        return self.latticeForwardProb

    # ------------------------------------------------------------------ pruning
    def _renumber(self):
        for idx, nd in enumerate(self.sortedNodes):
            nd.index = idx

    def prune(self, threshold=100.0):
        if threshold < 0.0:
            raise ValueError("Lattice pruning threshold < 0.0.")
        self.initial.removeLinks(threshold)
        for nd in self.nodes:
            if nd is not None:
                nd.removeLinks(threshold)
        self._clearSorted(); self._topoSort()
        gone = [nd.index for nd in self.nodes if nd is not None and nd.color == 0]
        for idx in gone:
            self.nodes[idx] = None
        for key in [k for k in sorted(self.final) if self.final[k].color == 0]:
            del self.final[key]
        self._renumber()

    def _edgeIterator(self):
        out = list(self.initial.iter())
        for nd in self.nodes:
            if nd is not None:
                out.extend(nd.iter())
        return out

    def pruneEdges(self, edgesN=0):
        edges = self._edgeIterator()
        if edgesN >= len(edges):
            return
        scores = sorted(e.gamma for e in edges)
        self.prune(scores[edgesN])

    def purge(self):
        self._setColor(0)
        self.initial.success = False
        for nd in self.nodes:
            if nd is not None:
                nd.success = False
        for nd in self.final.values():
            nd.success = False
        self._clearSorted()
        order = []
        # _purgeNode with the recursion unrolled: frame = [node, iterator, success so far]
        first = [self.initial, self.initial.iter(), self.initial.final]; self.initial.color = 1
        stack = [first]
        while stack:
            fr = stack[-1]
            nxt = next(fr[1], None)
            if nxt is None:
                nd = fr[0]; nd.success = fr[2]; nd.color = 2
                if fr[2]:
                    order.append(nd)
                stack.pop()
                if stack and fr[2]:
                    stack[-1][2] = True
            else:
                ch = nxt.next
                if ch.color == 2:
                    if ch.success:
                        fr[2] = True
                elif ch.color == 1:
                    raise ValueError("Node %d is gray; graph is not acyclic." % ch.index)
                else:
                    ch.color = 1; stack.append([ch, ch.iter(), ch.final])
        self.sortedNodes = order[::-1]
        gone = [nd.index for nd in self.nodes if nd is not None and not nd.success]
        for idx in gone:
            self.nodes[idx] = None
        for key in [k for k in sorted(self.final) if not self.final[k].success]:
            del self.final[key]
        self._renumber()

    # ------------------------------------------------------------------ files
    @staticmethod
    def _edge_text(e, writeData):
        s = "%10d  %10d  %10d  %10d" % (e.prev.index, e.next.index, e.input, e.output)
        s += "\n" if abs(float(e.cost)) < 1.0E-04 else "  %12g\n" % float(e.cost)
        if writeData:
            s += "%4d  %4d  %8.4f  %8.4f  %8.4f\n" % (e.start, e.end, e.ac, e.lm, e.gamma)
        return s

    def write(self, fileName, writeData=False):
        self._topoSort()
        out = []
        for nd in self.sortedNodes:
            if nd.final:
                continue
            for e in nd.iter():
                out.append(self._edge_text(e, writeData))
        for nd in self._finis():
            for e in nd.iter():
                out.append(self._edge_text(e, writeData))
            out.append("%10d\n" % nd.index if float(nd.cost) == 0.0 else "%10d  %12g\n" % (nd.index, float(nd.cost)))
        open(fileName, "w").write("".join(out))

    def _rows(self, symbols, byInput, htk, cfrom, frameInterval):
        tok = self._bestToken()
        endX = tok.edge.end; wscore = float(tok.ac)
        words = []; starts = []; durations = []; scores = []; lastPhoneX = 0
        while True:
            if byInput:
                phoneX = tok.edge.input
                if phoneX != 0 or phoneX != lastPhoneX:
                    startX = tok.edge.start
                    words.append(symbols[phoneX]); starts.append(cfrom + startX * frameInterval); durations.append((endX - startX) * frameInterval)
                    oscore = 0.0 if tok.prev is None else float(tok.prev.ac)
                    scores.append(wscore - oscore); endX = startX; lastPhoneX = phoneX; wscore = oscore
            else:
                outX = tok.edge.output
                if outX != 0:
                    startX = tok.edge.start
                    beg = cfrom + startX * frameInterval
                    if htk:
                        words.append(symbols[outX]); starts.append(beg); durations.append((endX - startX + 1) * frameInterval)
                        oscore = 0.0 if tok.prev is None else float(tok.prev.ac)
                        scores.append(wscore - oscore); endX = startX - 1; wscore = oscore
                    else:
                        ln = (endX - startX) * frameInterval
                        entry = symbols[outX]
                        while True:
                            colon = entry.find(":")
                            word = entry
                            if colon >= 0:
                                word = entry[colon + 1:]; ln /= 2; beg += ln
                            words.append(word); starts.append(beg); durations.append(ln)
                            oscore = 0.0 if tok.prev is None else float(tok.prev.ac)
                            scores.append(wscore - oscore); endX = startX; wscore = oscore
                            if colon >= 0:
                                entry = entry[:colon]; beg = cfrom + startX * frameInterval
                            else:
                                break
            tok = tok.prev
            if tok is None:
                break
        return words, starts, durations, scores

    def writeCTM(self, outSymbols, conv, channel, spk, utt, cfrom, score, fileName, frameInterval=0.01, endMarker="</s>", phones=False):
        w, s, d, sc = self._rows(outSymbols, phones, False, cfrom, frameInterval)
        with open(fileName, "a") as fp:
            fp.write(";; %s %10.4f %10.4f\n" % (utt, cfrom, score))
            for i in range(len(w) - 1, -1, -1):
                if w[i] == endMarker:
                    continue
                fp.write("%s %s %7.2f %7.2f %-20s %7.2f\n" % (conv, channel, s[i], d[i], w[i], sc[i]))

    def writeHypoHTK(self, outSymbols, conv, channel, spk, utt, cfrom, score, fileName, flag=0, frameInterval=0.01, endMarker="</s>"):
        w, s, d, sc = self._rows(outSymbols, False, True, cfrom, frameInterval)
        with open(fileName, "a") as fp:
            fp.write("\"%s.rec\"\n" % utt)
            for i in range(len(w) - 1, -1, -1):
                if w[i] == endMarker:
                    continue
                if flag & 1:
                    fp.write("%d %d " % (int(s[i] * 10e7), int((s[i] + d[i]) * 10e7)))
                fp.write("%s" % w[i])
                if flag & 2:
                    fp.write(" %f" % sc[i])
                fp.write("\n")
            fp.write(".\n")

    def writeWordConfs(self, outSymbols, fileName, uttId, endMarker="</s>"):
        tok = self._bestToken(); words = []; gammas = []
        while tok is not None:
            if tok.edge.output != 0:
                words.append(outSymbols[tok.edge.output]); gammas.append(tok.edge.gamma)
            tok = tok.prev
        output = uttId
        for i in range(len(words) - 1, -1, -1):
            g = math.exp(-gammas[i])
            if g < 1.0E-04:
                g = 0.0
            elif g > 1.0:
                g = 1.0
            if words[i] == endMarker:
                continue
            output += " { {%s} %8.6f}" % (words[i], g)
        with open(fileName, "a") as fp:
            fp.write(output + "\n")

    # ------------------------------------------------------------------ reporting (tests)
    def state(self):
        """per edge (creation order): gamma, on its node's list; per node (creation order): index, held by the lattice, fwd, bwd"""
        E = len(self.allEdges); n = len(self.allMade)
        gamma = np.array([e.gamma for e in self.allEdges], np.float64)
        live = np.zeros(E, np.int32)
        for nd in self.allMade:
            for e in nd.iter():
                live[e.serial] = 1
        held = np.zeros(n, np.int32)
        held[self.initial.serial] = 1
        for nd in self.nodes:
            if nd is not None:
                held[nd.serial] = 1
        for nd in self.final.values():
            held[nd.serial] = 1
        return dict(gamma=gamma, edgeLive=live, nodeIndex=np.array([nd.index for nd in self.allMade], np.int32), nodeLive=held,
                    fwd=np.array([nd.fwd for nd in self.allMade]), bwd=np.array([nd.bwd for nd in self.allMade]))
