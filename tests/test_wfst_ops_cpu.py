"""WFSTFlyWeight::reverse / reverseRead / write(useSymbols) (asr/decoder/wfstFlyWeight.cc:141-297, 415-463, 499-516) and the small boundary
methods of SURVEY 8(b), through the C-ABI, against oracle/oracle_wfst.py (an object-graph restatement shaped like the reference's container).
No GPU: these are host operations (dsr_wfst_* need no device).  Files are compared byte for byte."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_wfst as OW
from tests import synth


def _write_text_graph(path, arcs, finals, symbolic=None):
    """AT&T text: arcs first (first arc's source = initial state), then the final states"""
    with open(path, "w") as f:
        for (s1, s2, i, o, c) in arcs:
            if symbolic:
                st, il, ol = symbolic
                f.write("%s %s %s %s" % (st[s1], st[s2], il[i], ol[o]))
            else:
                f.write("%d %d %d %d" % (s1, s2, i, o))
            f.write("\n" if c == 0.0 else " %r\n" % float(np.float32(c)))
        for (s, c) in finals:
            f.write(("%s" % (symbolic[0][s] if symbolic else s)) + ("\n" if c == 0.0 else " %r\n" % float(np.float32(c))))


def _graphs(seed, S=60, ties=False):
    arcs, fin = synth.random_wfst(S, 12, seed=seed, outdeg=3, eps_frac=0.15, out_frac=0.3, nWords=9, nFinal=6, ties=ties)
    rng = np.random.default_rng(seed)
    arcs = list(arcs)
    arcs.insert(3, (5, 5, 0, 4, 1.5))          # an epsilon-input self loop WITH an output: kept by read, dropped by reverseRead
    arcs.insert(7, (9, 9, 0, 0, 0.25))         # epsilon:epsilon self loop: dropped by both
    arcs.insert(9, (arcs[0][0], arcs[0][0], 3, 0, 0.0))   # a self loop on the initial state, zero cost
    arcs.append((11, 12, 2, 1, 0.00005))       # a cost below MinimumCost: left out by the symbolic writer only
    fin = list(fin) + [(arcs[0][0], 0.75)] if rng.random() < 2 else fin    # the initial state's index also as a final state (a node of its own)
    return arcs, fin


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_reverse_and_reverse_read_match_the_restatement(tmp_path, seed):
    import dsr._capi as K
    L = K.load()
    arcs, fin = _graphs(seed)
    src = str(tmp_path / "g.txt"); _write_text_graph(src, arcs, fin)
    # ---- oracle
    og = OW.FlyWeight(); og.read_text(src)
    orev = OW.FlyWeight(); orev.reverse(og)
    orr = OW.FlyWeight(); orr.reverse_read(src)
    # ---- product
    g = K.Wfst(); g.read(src, False)
    rev = K.Wfst(); K.check(L.dsr_wfst_reverse(rev.h, g.h))
    rr = K.Wfst(); K.check(L.dsr_wfst_reverse_read(rr.h, src.encode()))
    for binary in (False, True):
        for (po, pp, name) in ((og, g, "g"), (orev, rev, "rev"), (orr, rr, "rr")):
            fo = str(tmp_path / ("o_%s_%d" % (name, binary))); fp = str(tmp_path / ("p_%s_%d" % (name, binary)))
            po.write(fo, binary); pp.write(fp, binary)
            assert open(fo, "rb").read() == open(fp, "rb").read(), (name, binary)
    # the reversed graph: one final state (the source's initial state), a super-initial node with one epsilon arc per final state of the source
    ex = rev.export()
    assert int(ex["nodeState"][0]) == OW.MAXIMUM_INDEX - 3 and int(ex["nodeFinal"].sum()) == 1
    nfin = len(set(s for s, _ in fin))
    assert int(ex["arcOff"][1] - ex["arcOff"][0]) == nfin and not ex["arcIn"][:nfin].any()
    # reversing twice gives back every original arc (plus the two layers of super nodes)
    rev2 = K.Wfst(); K.check(L.dsr_wfst_reverse(rev2.h, rev.h))
    assert L.dsr_wfst_num_arcs(rev2.h) == L.dsr_wfst_num_arcs(g.h) + nfin + 1
    # a final-state line before any arc mentions the state: find() without create -> jkey_error
    bad = str(tmp_path / "bad.txt"); open(bad, "w").write("7\n1 2 3 0\n")
    gb = K.Wfst()
    with pytest.raises(K.DsrError) as e:
        K.check(L.dsr_wfst_reverse_read(gb.h, bad.encode()))
    assert "No state 7 exists." in str(e.value)
    with pytest.raises(OW.KeyErrorJ):
        OW.FlyWeight().reverse_read(bad)


@pytest.mark.parametrize("with_state_lexicon", [False, True])
def test_write_with_symbols(tmp_path, with_state_lexicon):
    import dsr._capi as K
    L = K.load()
    arcs, fin = _graphs(5, S=30)
    nS = 1 + max(max(a[0], a[1]) for a in arcs)
    st = ["st%03d" % i for i in range(nS)]; il = ["<eps>"] + ["d%d" % i for i in range(1, 13)]; ol = ["<eps>"] + ["w%d" % i for i in range(1, 10)]
    src = str(tmp_path / "g.txt"); _write_text_graph(src, arcs, fin, symbolic=(st, il, ol) if with_state_lexicon else None)
    if not with_state_lexicon:
        _write_text_graph(src, arcs, fin)

    def lexfile(name, syms):
        p = str(tmp_path / name)
        with open(p, "w") as f:
            for i, s_ in enumerate(syms):
                f.write("%30s %10d\n" % (s_, i))
        lx = K.vp(); K.check(L.dsr_lexicon_create(name.encode(), p.encode(), C.byref(lx))); return lx
    lxs = lexfile("states", st) if with_state_lexicon else None
    lxi, lxo = lexfile("in", il), lexfile("out", ol)
    g = K.Wfst(); K.check(L.dsr_wfst_set_lexicons(g.h, lxs, lxi, lxo)); g.read(src, False)
    og = OW.FlyWeight(st if with_state_lexicon else None, il, ol); og.read_text(src)
    for binary in (False, True):
        fo = str(tmp_path / ("o%d" % binary)); fp = str(tmp_path / ("p%d" % binary))
        og.write(fo, binary, use_symbols=True)
        K.check(L.dsr_wfst_write_symbols(g.h, fp.encode(), int(binary), 1))
        assert open(fo, "rb").read() == open(fp, "rb").read(), binary
    txt = open(str(tmp_path / "p0")).read()
    assert ("st005" in txt) == with_state_lexicon and "w4" in txt and "5e-05" not in txt       # the 5e-5 cost is below MinimumCost: not written
    # a symbolic file reads back into the same graph (numeric dump identical)
    g2 = K.Wfst(); K.check(L.dsr_wfst_set_lexicons(g2.h, lxs, lxi, lxo)); g2.read(str(tmp_path / "p0"), False)
    a, b = str(tmp_path / "n1"), str(tmp_path / "n2")
    if with_state_lexicon:                                                   # (without it the final-state lines are numeric and so are the states)
        g.write(a, False); g2.write(b, False)
        la = [l.split() for l in open(a)]; lb = [l.split() for l in open(b)]
        assert sorted(x[:4] for x in la) == sorted(x[:4] for x in lb)          # (a file read back lists a node's arcs in reverse: arcs are prepended)
    # no lexica: DSR_E_KEY
    g3 = K.Wfst(); g3.add_arc(0, 1, 2, 3, 1.0)
    with pytest.raises(K.DsrError):
        K.check(L.dsr_wfst_write_symbols(g3.h, str(tmp_path / "x").encode(), 0, 1))
    for lx in (lxs, lxi, lxo):
        if lx:
            L.dsr_lexicon_destroy(lx)


def test_python_face_of_the_graph_methods(tmp_path):
    """WFSTFlyWeightPtr.reverse / reverseRead / write(useSymbols) (decoder.i:52-70) and DecoderFlyWeightPtr.writeCTM / setTokenMemoryLimit"""
    from dsr.asr import decoder as D, dictionary as Dict
    arcs, fin = _graphs(8, S=20)
    src = str(tmp_path / "g.txt"); _write_text_graph(src, arcs, fin)
    w = D.WFSTFlyWeightPtr(); w.read(src)
    r = D.WFSTFlyWeightPtr(); r.reverse(w)
    r2 = D.WFSTFlyWeightPtr(); r2.reverseRead(src)
    og = OW.FlyWeight(); og.read_text(src); orev = OW.FlyWeight(); orev.reverse(og); orr = OW.FlyWeight(); orr.reverse_read(src)
    for (po, pp, nm) in ((orev, r, "a"), (orr, r2, "b")):
        fo, fp = str(tmp_path / ("o" + nm)), str(tmp_path / ("p" + nm))
        po.write(fo, False); pp.write(fp, False)
        assert open(fo, "rb").read() == open(fp, "rb").read()


def test_sample_feature_read_through_the_boundary(tmp_path):
    """dsr_sample_feature_read (feature.cc:243-393) against an independent reading of the same RIFF files (Python's wave module): 8/16/24/32-bit PCM, channel
    pick, sample ranges, norm, the reference's error branches in the reference's order.  Host only (the samples are held until the first next())."""
    import wave
    import dsr._capi as K
    L = K.load()
    rng = np.random.default_rng(3)

    def mk(path, sw, nch, n, rate=16000):
        if sw == 1:
            a = rng.integers(0, 256, (n, nch)).astype(np.uint8); raw = a.tobytes(); val = a.astype(np.int64) - 128
        elif sw == 2:
            a = rng.integers(-32768, 32768, (n, nch)).astype("<i2"); raw = a.tobytes(); val = a.astype(np.int64)
        elif sw == 3:
            a = rng.integers(-(1 << 23), 1 << 23, (n, nch)).astype(np.int64); raw = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in a.reshape(-1)); val = a
        else:
            a = rng.integers(-(1 << 31), 1 << 31, (n, nch)).astype("<i4"); raw = a.tobytes(); val = a.astype(np.int64)
        w = wave.open(path, "wb"); w.setnchannels(nch); w.setsampwidth(sw); w.setframerate(rate); w.writeframes(raw); w.close()
        return val

    def read(fn, chX=1, cfrom=0, to=-1, norm=0.0, outrate=-1):
        h = K.vp(); K.check(L.dsr_sample_feature_create(320, 160, 0, b"Sample", C.byref(h)))
        n = C.c_int(0)
        st = L.dsr_sample_feature_read(h, fn.encode(), 0, 16000, chX, 1, cfrom, to, outrate, C.c_float(norm), C.byref(n))
        rate = L.dsr_sample_feature_sample_rate(h); L.dsr_stream_release(h)
        return st, n.value, rate
    for sw in (1, 2, 3, 4):
        p = str(tmp_path / ("a%d.wav" % sw)); val = mk(p, sw, 3, 1000, rate=8000 if sw == 3 else 16000)
        st, n, rate = read(p, chX=2); assert st == 0 and n == 1000 and rate == (8000 if sw == 3 else 16000)
        st, n, _ = read(p, chX=3, cfrom=100, to=499); assert st == 0 and n == 400
        st, n, _ = read(p, chX=1, cfrom=990, to=5000); assert st == 0 and n == 10            # `to` beyond the file: clipped to the last frame
        assert read(p, chX=0)[0] == 4 and read(p, chX=4)[0] == 4                       # jconsistency_error (DSR_E_CONSISTENCY)
        assert read(p, cfrom=600, to=500)[0] == 8 and read(p, chX=0, cfrom=600, to=500)[0] == 8   # jio_error first: the range is checked before the channel
        assert read(p, outrate=44100)[0] != 0
    assert read(str(tmp_path / "missing.wav"))[0] == 8
    open(str(tmp_path / "junk.wav"), "wb").write(b"RIFFxxxxJUNK")
    assert read(str(tmp_path / "junk.wav"))[0] == 8
