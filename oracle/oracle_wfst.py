"""oracle/oracle_wfst.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ may import it).

CPU restatement of the WFSTFlyWeight container operations the decoder's graph API exposes beyond read/write
(reference: asr/decoder/wfstFlyWeight.cc, asr/decoder/wfstFlyWeight.h), on an object graph shaped like the reference's:
a node object per state, `_initial` a node of its own, `_nodes` and `_final` ordered maps state -> node, every node a
singly linked edge list that new edges are PREPENDED to (Node::_addEdgeForce, :552-556).

    _addFinal        wfstFlyWeight.cc:72-92
    find             :94-118
    _readText        :299-365
    reverse          :141-213
    reverseRead      :215-297
    write            :415-463   Edge::write (numeric) :474-495, Edge::write (symbols) :499-516, Node::write :558-575

Parity unpinned: the reference holds no files written by these calls; restated from the source text.  The product is compared
with this restatement file for file, byte for byte (tests/test_wfst_ops_cpu.py).
"""
import struct

MAXIMUM_INDEX = 536870911          # WFSTFlyWeight::Node::_MaximumIndex (:522)
END_MARKER = 2147483647            # WFSTFlyWeight::EndMarker (:32)


class KeyErrorJ(Exception):
    """jkey_error"""


class ConsistencyErrorJ(Exception):
    """jconsistency_error"""


class IOErrorJ(Exception):
    """jio_error"""


class Edge(object):
    __slots__ = ("prev", "next", "input", "output", "cost", "link")

    def __init__(self, frm, to, inp, out, cost=0.0):
        self.prev, self.next, self.input, self.output, self.cost, self.link = frm, to, inp, out, _f32(cost), None


class Node(object):
    __slots__ = ("index", "cost", "final", "edges")

    def __init__(self, idx, cost=0.0):
        self.index, self.cost, self.final, self.edges = idx, 0.0, False, None

    def add_edge_force(self, e):
        e.link = self.edges; self.edges = e

    def iter_edges(self):
        e = self.edges
        while e is not None:
            yield e
            e = e.link


def _f32(x):
    return struct.unpack("f", struct.pack("f", float(x)))[0]


def _g12(x):
    """printf("%12g", x)"""
    return "%12g" % x


class SortedOutputNode(Node):
    """WFSTFlyWeightSortedOutput::Node::_addEdgeForce (wfstFlyWeight.cc:754-776): arcs ordered by (output, input), a new one before the first not smaller"""
    __slots__ = ()

    def add_edge_force(self, e):
        ptr = self.edges; old = ptr
        while ptr is not None and ptr.output < e.output:
            old = ptr; ptr = ptr.link
        while ptr is not None and ptr.output == e.output and ptr.input < e.input:
            old = ptr; ptr = ptr.link
        if ptr is old:
            e.link = self.edges; self.edges = e
        else:
            e.link = ptr; old.link = e


class FlyWeight(object):
    NodeT = Node

    def __init__(self, statelex=None, inlex=None, outlex=None):
        """lexica: lists of symbols (index = position) or None"""
        self.statelex, self.inlex, self.outlex = statelex, inlex, outlex
        self._clear()

    def _clear(self):
        self.initial = None; self.nodes = {}; self.final = {}

    # ---- container
    def _add_final(self, state, cost):
        if state in self.final:
            raise ConsistencyErrorJ("Automaton already has final node %d." % state)
        nd = self.nodes.pop(state, None)
        if nd is None:
            nd = self.NodeT(state)
        nd.cost = _f32(cost); nd.final = True
        self.final[state] = nd

    def find(self, state, create=False):
        if self.initial.index == state:
            return self.initial
        if state in self.nodes:
            return self.nodes[state]
        if state in self.final:
            return self.final[state]
        if not create:
            raise KeyErrorJ("No state %u exists." % state)
        self.nodes[state] = self.NodeT(state)
        return self.nodes[state]

    def _field(self, lex, tok):
        """strtoul(tok, &p, 0); the lexicon when no digits were consumed"""
        t = tok.strip(); s = t; neg = False
        if s[:1] in "+-":
            neg = s[0] == "-"; s = s[1:]
        digits = ""
        if s[:2].lower() == "0x" and len(s) > 2 and s[2] in "0123456789abcdefABCDEF":
            i = 2
            while i < len(s) and s[i] in "0123456789abcdefABCDEF":
                i += 1
            v = int(s[2:i], 16); digits = s[:i]
        elif s[:1] == "0":
            i = 1
            while i < len(s) and s[i] in "01234567":
                i += 1
            v = int(s[:i], 8); digits = s[:i]
        else:
            i = 0
            while i < len(s) and s[i].isdigit():
                i += 1
            if i > 0:
                v = int(s[:i]); digits = s[:i]
        if digits == "":
            if lex is None:
                raise KeyErrorJ(tok)
            return lex.index(tok)
        return (-v if neg else v) & 0xFFFFFFFF

    def read_text(self, path):
        self._clear()
        for line in open(path):
            tok = line.split()[:6]
            if not tok:
                continue
            s1 = self._field(self.statelex, tok[0])
            n = min(len(tok), 5)                                      # (the tokeniser stops at six tokens and counts at most five)
            if n == 1:
                self._add_final(s1, 0.0)
            elif n == 2:
                self._add_final(s1, _scanf_f(tok[1]))
            elif n == 4 or n == 5:
                s2 = self._field(self.statelex, tok[1])
                if self.initial is None:
                    self.initial = frm = self.NodeT(s1)
                else:
                    frm = self.find(s1, True)
                to = self.find(s2, True)
                inp = self._field(self.inlex, tok[2]); out = self._field(self.outlex, tok[3])
                if s1 == s2 and inp == 0 and out == 0:
                    continue
                cost = _scanf_f(tok[4]) if n == 5 else 0.0
                frm.add_edge_force(Edge(frm, to, inp, out, cost))
            else:
                raise IOErrorJ("Transducer file is inconsistent.")

    def reverse(self, wfst):
        self._clear()
        self.initial = rinitial = self.NodeT(MAXIMUM_INDEX - 3)
        self._add_final(wfst.initial.index, 0.0)
        rfinal = self.find(wfst.initial.index)
        for e in wfst.initial.iter_edges():                          # from the final (i.e. initial) node
            r2 = self.find(e.next.index, True)
            r2.add_edge_force(Edge(r2, rfinal, e.input, e.output, e.cost))
        for st in sorted(wfst.final):                                # from the super initial node
            nd = wfst.final[st]; rn = self.find(nd.index, True)
            rinitial.add_edge_force(Edge(rinitial, rn, 0, 0, nd.cost))
        for mp in (wfst.final, wfst.nodes):                          # from the final nodes, then from the internal nodes
            for st in sorted(mp):
                n1 = mp[st]; r1 = self.find(n1.index, True)
                for e in n1.iter_edges():
                    r2 = self.find(e.next.index, True)
                    r2.add_edge_force(Edge(r2, r1, e.input, e.output, e.cost))

    def reverse_read(self, path):
        self._clear()
        self.initial = rinitial = self.NodeT(MAXIMUM_INDEX - 3)
        initial_flag = False
        for line in open(path):
            tok = line.split()[:6]
            if not tok:
                continue
            n = min(len(tok), 5)
            s1 = self._field(self.statelex, tok[0])
            if n == 1:
                rn = self.find(s1); rinitial.add_edge_force(Edge(rinitial, rn, 0, 0))
            elif n == 2:
                rn = self.find(s1); rinitial.add_edge_force(Edge(rinitial, rn, 0, 0, _scanf_f(tok[1])))
            elif n == 4 or n == 5:
                s2 = self._field(self.statelex, tok[1])
                if not initial_flag:
                    self._add_final(s1, 0.0); initial_flag = True
                frm = self.find(s1, True); to = self.find(s2, True)
                inp = self._field(self.inlex, tok[2]); out = self._field(self.outlex, tok[3])
                if s1 == s2 and inp == 0:
                    continue
                cost = _scanf_f(tok[4]) if n == 5 else 0.0
                to.add_edge_force(Edge(to, frm, inp, out, cost))
            else:
                raise IOErrorJ("Transducer file %s is inconsistent." % path)

    # ---- writers
    def _edge_bytes(self, e, binary, use_symbols):
        if use_symbols:
            ins, outs = self.inlex[e.input], self.outlex[e.output]
            if not self.statelex:
                s = "%10d  %10d  %10s  %20s" % (e.prev.index, e.next.index, ins, outs)
            else:
                s = "%25s  %25s  %10s  %20s" % (self.statelex[e.prev.index], self.statelex[e.next.index], ins, outs)
            s += "\n" if abs(e.cost) < 1.0E-04 else "  %s\n" % _g12(e.cost)
            return s.encode()
        if binary:
            return struct.pack(">iiiiifi", 6, e.prev.index, e.next.index, _i32(e.input), _i32(e.output), e.cost, END_MARKER)
        s = "%10d  %10d  %10d  %10d" % (e.prev.index, e.next.index, _i32(e.input), _i32(e.output))
        s += "\n" if e.cost == 0.0 else "  %s\n" % _g12(e.cost)
        return s.encode()

    def _node_bytes(self, nd, binary):
        if binary:
            return struct.pack(">iifi", 3, nd.index, nd.cost, END_MARKER)
        return ("%10d\n" % nd.index if nd.cost == 0.0 else "%10d  %s\n" % (nd.index, _g12(nd.cost))).encode()

    def write(self, path, binary=True, use_symbols=False):
        out = []
        for e in self.initial.iter_edges():
            out.append(self._edge_bytes(e, binary, use_symbols))
        for st in sorted(self.nodes):
            for e in self.nodes[st].iter_edges():
                out.append(self._edge_bytes(e, binary, use_symbols))
        for st in sorted(self.final):
            nd = self.final[st]
            for e in nd.iter_edges():
                out.append(self._edge_bytes(e, binary, use_symbols))
            out.append(self._node_bytes(nd, binary))
        if binary:
            out.append(struct.pack(">i", END_MARKER))
        with open(path, "wb") as f:
            f.write(b"".join(out))


def _i32(u):
    return u - (1 << 32) if u >= (1 << 31) else u


def _scanf_f(tok):
    """sscanf(tok, "%f", &cost)"""
    return _f32(float(tok))


class FlyWeightSortedOutput(FlyWeight):
    """WFSTFlyWeightSortedOutput (wfstFlyWeight.h:403-424)"""
    NodeT = SortedOutputNode

    def add_arc(self, s1, s2, inp, out, cost=0.0):
        """what _readText does for one arc line (wfstFlyWeight.cc:325-358)"""
        if self.initial is None:
            self.initial = frm = self.NodeT(s1)
        else:
            frm = self.find(s1, True)
        to = self.find(s2, True)
        if s1 == s2 and inp == 0 and out == 0:
            return
        frm.add_edge_force(Edge(frm, to, inp, out, cost))

    def add_final(self, s, cost=0.0):
        self._add_final(s, cost)
