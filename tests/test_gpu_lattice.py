"""Lattice generation and its gather (row f2) against the oracle's restatement of _Decoder::lattice / _majorTrace /
_minorTrace (asr/decoder/decoder.h:805-953) over the 'worse' chains of _placeOnList (:531-541): node numbering, edge list (creation order),
frames, ac/lm doubles -- all bit for bit -- and the bytes of Lattice::write (asr/lattice/lattice.cc:715-757)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _graphs(dsr, oracle, arcs, fin):
    go, gd = oracle.Wfst(), dsr.Wfst()
    for a in arcs:
        go.add_arc(*a); gd.add_arc(*a)
    for s, c in fin:
        go.add_final(s, c); gd.add_final(s, c)
    return go, gd


def _same_lattice(Lo, Ld):
    for k in ("nodeFinal", "from", "to", "in", "out", "start", "end"):
        assert np.array_equal(Lo[k], Ld[k]), k
    for k in ("ac", "lm"):
        assert np.array_equal(Lo[k].view(np.int64), Ld[k].view(np.int64)), k         # doubles, bit for bit


@pytest.mark.parametrize("seed,S,nDist,T,beam,kw", [
    (1, 300, 16, 40, 1e9, dict()),
    (2, 1000, 64, 80, 30.0, dict()),
    (3, 2000, 128, 60, 12.0, dict(lmPenalty=0.7)),
    (4, 2000, 128, 60, 8.0, dict(silPenalty=1.5, silenceX=3)),
    (5, 500, 32, 40, 50.0, dict(ties=True)),
    (6, 300, 8, 40, 25.0, dict(eps_frac=0.35, out_frac=0.3)),
    (7, 400, 16, 30, 40.0, dict(nFinal=0)),
    (8, 300, 8, 30, 30.0, dict(eps_frac=0.2, silPenalty=0.9, silenceX=0, lmPenalty=0.3)),     # the silence symbol on the epsilon arcs (decoder.h:975-977)
])
def test_lattice_matches_oracle(dsr, oracle, cuda, tmp_path, seed, S, nDist, T, beam, kw):
    import torch
    gkw = {k: kw[k] for k in ("ties", "eps_frac", "nFinal", "out_frac") if k in kw}
    dkw = {k: kw[k] for k in ("lmPenalty", "silPenalty", "silenceX") if k in kw}
    arcs, fin = synth.random_wfst(S, nDist, seed=seed, nWords=200, **gkw)
    go, gd = _graphs(dsr, oracle, arcs, fin)
    rng = np.random.default_rng(200 + seed)
    sc = rng.uniform(0, 10, (3, T, nDist)).astype(np.float32)
    if kw.get("ties"):
        sc = np.round(sc)
    nfr = [T, T - 7, 2]
    dec = dsr.Decoder(beam=beam, lmScale=12.0, maxActive=8192, streams=2, latticeTokens=400000, **dkw)
    dec.set(gd)
    out = dec.decode_batch(torch.from_numpy(sc).to(cuda), torch.tensor(nfr, dtype=torch.int32, device=cuda))
    plain = dsr.Decoder(beam=beam, lmScale=12.0, maxActive=8192, streams=2, **dkw); plain.set(gd)
    out0 = plain.decode_batch(torch.from_numpy(sc).to(cuda), torch.tensor(nfr, dtype=torch.int32, device=cuda))
    with pytest.raises(dsr.DsrError) as e:
        plain.lattice(0)
    assert e.value.status == 4                                            # jconsistency_error "Must enable lattice generation during decoding."
    for u in range(3):
        assert out[u]["status"] == 0 and out[u]["score"] == out0[u]["score"] and np.array_equal(out[u]["arcs"], out0[u]["arcs"])     # the bookkeeping does not move the 1-best
        fo = str(tmp_path / ("o%d.lat" % u))
        try:
            ro = go.decode(sc[u, :nfr[u]], beam=beam, lmScale=12.0, lattice=True, eosX=7, latticeFile=fo, writeData=True, **dkw); cyclic = False
        except ValueError:
            ro = go.decode(sc[u, :nfr[u]], beam=beam, lmScale=12.0, lattice=True, eosX=7, **dkw); cyclic = True
        assert ro["rc"] == 0 and ro["score"] == out[u]["score"]
        L = dec.lattice(u, eosX=7)
        assert L.finalStatesN == ro["finalStatesN"]
        _same_lattice(ro["lattice"], L.data)
        assert len(L.data["from"]) >= nfr[u] - 1
        fd = str(tmp_path / ("d%d.lat" % u))
        if cyclic:                                                        # (a self loop taken in the last frame + an epsilon arc into a final state:
            with pytest.raises(dsr.DsrError) as e2:                       #  the reference's (state, frame) keys collide and its _topoSort throws)
                L.write(fd, writeData=True)
            assert e2.value.status == 4
        else:
            L.write(fd, writeData=True)
            assert open(fd, "rb").read() == open(fo, "rb").read()         # Lattice::write, byte for byte
        # the flat image used by the gather
        L2 = dsr.Lattice.unpack(L.pack())
        _same_lattice(L.data, L2.data)


@pytest.mark.parametrize("seed,S,nDist,T,beam,kw", [
    (2, 1000, 64, 80, 30.0, dict()),
    (3, 2000, 128, 60, 12.0, dict(lmPenalty=0.7)),
    (4, 2000, 128, 60, 8.0, dict(silPenalty=1.5, silenceX=3)),
    (6, 300, 8, 40, 25.0, dict(eps_frac=0.35, out_frac=0.3)),
])
def test_lattice_operations_on_a_decoded_lattice(dsr, oracle, cuda, tmp_path, seed, S, nDist, T, beam, kw):
    """asr/lattice on what the search hands over (lattice.cc:122-379, 648-757): rescoring the decoder's lattice with the decoder's own scale and
    penalties finds the decoder's 1-best again (words and, within the float accumulation of two different sums, its score); posteriors, pruning and the
    files agree with the object-graph restatement (oracle/oracle_lattice.py) run on the ORACLE decoder's lattice, bit for bit / byte for byte."""
    import importlib, sys, os
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    olat = importlib.import_module("oracle_lattice")
    gkw = {k: kw[k] for k in ("ties", "eps_frac", "nFinal", "out_frac") if k in kw}
    dkw = {k: kw[k] for k in ("lmPenalty", "silPenalty", "silenceX") if k in kw}
    arcs, fin = synth.random_wfst(S, nDist, seed=seed, nWords=200, **gkw)
    go, gd = _graphs(dsr, oracle, arcs, fin)
    rng = np.random.default_rng(200 + seed)
    sc = rng.uniform(0, 10, (2, T, nDist)).astype(np.float32)
    nfr = [T, T - 9]
    dec = dsr.Decoder(beam=beam, lmScale=12.0, maxActive=8192, streams=2, latticeTokens=400000, **dkw)
    dec.set(gd)
    out = dec.decode_batch(torch.from_numpy(sc).to(cuda), torch.tensor(nfr, dtype=torch.int32, device=cuda))
    P = dict(lmScale=12.0, lmPenalty=dkw.get("lmPenalty", 0.0), silPenalty=dkw.get("silPenalty", 0.0), silenceX=dkw.get("silenceX", 0xFFFFFFFF) & 0xFFFFFFFF)
    for u in range(2):
        ro = go.decode(sc[u, :nfr[u]], beam=beam, lmScale=12.0, lattice=True, eosX=7, **dkw)
        L = dec.lattice(u, eosX=7); O = olat.Lattice.from_arrays(ro["lattice"])
        try:
            s = L.rescore(**P)
        except dsr.DsrError as e:                                          # the (state, frame) collision of the epsilon case: not a DAG, both sides refuse
            assert e.status == 4
            with pytest.raises(ValueError):
                O.rescore(**P)
            continue
        so = O.rescore(**P)
        assert np.float32(s).view(np.int32) == np.float32(so).view(np.int32)
        hyp = list(L.bestHypo()); assert hyp == list(O.bestHypo()) and list(L.bestHypo(True)) == list(O.bestHypo(True))
        if out[u]["reachedFinal"]:
            # the search's own best path is in the lattice and wins the rescoring with the search's own parameters
            assert hyp == [int(w) for w in out[u]["words"] if w != 0]
            if not dkw.get("silPenalty"):
                assert abs(float(s) - out[u]["score"]) <= 2e-4 * abs(out[u]["score"])
        g = L.gammaProbs(acScale=1.0, **P); g2 = O.gammaProbs(acScale=1.0, **P)     # (acScale stays behind for the next rescore, lattice.cc:124)
        assert g == g2
        sa, sb = L.state(), O.state()
        for k in ("gamma", "fwd", "bwd"):
            assert np.array_equal(sa[k].view(np.int64), sb[k].view(np.int64)), k
        live = sa["edgeLive"] == 1
        assert np.all(sa["gamma"][live] >= 0.0) and np.any(sa["gamma"][live] < 1e-3)          # posteriors are probabilities; the best path's links are near 1
        L.prune(3.0); O.prune(3.0)                                                                 # links below a posterior of e^-3 go
        sa, sb = L.state(), O.state()
        for k in ("edgeLive", "nodeIndex", "nodeLive"):
            assert np.array_equal(sa[k], sb[k]), k
        assert sa["edgeLive"].sum() <= live.sum()                                              # (a narrow beam leaves one path: nothing to prune)
        s2 = L.rescore(**P); so2 = O.rescore(**P)
        assert np.float32(s2).view(np.int32) == np.float32(so2).view(np.int32) and list(L.bestHypo()) == list(O.bestHypo())
        assert np.float32(s2) == np.float32(s) and list(L.bestHypo()) == hyp                    # pruning by posterior keeps the best path
        a, b = str(tmp_path / ("p%d.lat" % u)), str(tmp_path / ("o%d.lat" % u))
        L.write(a, writeData=True); O.write(b, writeData=True)
        assert open(a, "rb").read() == open(b, "rb").read()


def test_lattice_epsilon_free_contains_the_best_path(dsr, oracle, cuda):
    """without epsilon arcs the lattice is exact: its best path has the decode score (property, no oracle needed)"""
    import torch
    arcs, fin = synth.random_wfst(800, 32, seed=31, eps_frac=0.0, out_frac=0.2, nWords=100)
    gd = dsr.Wfst()
    for a in arcs:
        gd.add_arc(*a)
    for s, c in fin:
        gd.add_final(s, c)
    rng = np.random.default_rng(31)
    T, lmS, pen = 50, 3.0, 0.4
    sc = rng.uniform(0, 10, (2, T, 32)).astype(np.float32)
    dec = dsr.Decoder(beam=25.0, lmScale=lmS, lmPenalty=pen, maxActive=8192, streams=2, latticeTokens=600000); dec.set(gd)
    out = dec.decode_batch(torch.from_numpy(sc).to(cuda))
    for u in range(2):
        L = dec.lattice(u).data
        n = len(L["nodeFinal"]); best = np.full(n, np.inf); best[0] = 0.0
        order = np.argsort(L["start"], kind="stable")
        w = L["ac"] + lmS * (L["lm"] + np.where(L["out"] != 0, pen, 0.0))
        for _ in range(T + 2):
            ch = False
            for e in order:
                v = best[L["from"][e]] + w[e]
                if v < best[L["to"][e]] - 1e-12:
                    best[L["to"][e]] = v; ch = True
            if not ch:
                break
        fb = min(best[i] for i in range(n) if L["nodeFinal"][i] == 1)
        assert abs(fb - out[u]["score"]) <= 1e-4 * abs(out[u]["score"])


def test_lattice_capacity_and_errors(dsr, cuda):
    import torch
    arcs, fin = synth.random_wfst(300, 16, seed=3)
    gd = dsr.Wfst()
    for a in arcs:
        gd.add_arc(*a)
    for s, c in fin:
        gd.add_final(s, c)
    sc = np.random.default_rng(0).uniform(0, 10, (1, 30, 16)).astype(np.float32)
    dec = dsr.Decoder(beam=1e9, lmScale=12.0, maxActive=8192, streams=1, latticeTokens=500); dec.set(gd)     # too small on purpose
    out = dec.decode_batch(torch.from_numpy(sc).to(cuda))
    assert out[0]["status"] == 2                                          # JALLOCATION: the placement log is full -- reported, never silent
    with pytest.raises(dsr.DsrError):
        dec.lattice(0)
    with pytest.raises(dsr.DsrError):
        dec.lattice(5)


@pytest.mark.parametrize("seed,S,nDist,T,beam,kw", [
    (11, 300, 16, 40, 1e9, dict()),
    (12, 1000, 32, 60, 30.0, dict(lmPenalty=0.7)),
    (13, 300, 8, 40, 25.0, dict(eps_frac=0.35, out_frac=0.3)),                       # epsilon tokens on the best path: skipped as "null arcs"
    (14, 400, 16, 30, 40.0, dict(nFinal=0)),                                          # no final state reached: the best token of _current
    (15, 300, 12, 50, 30.0, dict(eps_frac=0.2, silPenalty=0.9, silSym=3)),
])
def test_write_gmm(dsr, oracle, cuda, tmp_path, seed, S, nDist, T, beam, kw):
    """_Decoder::writeGMM (decoder.h:1018-1102): label runs of the 1-best path, the text file byte for byte against the oracle's rows"""
    import torch
    import ctypes as C
    from dsr.asr.dictionary import LexiconPtr
    from dsr.asr.decoder import WFSTFlyWeightPtr
    gkw = {k: kw[k] for k in ("eps_frac", "nFinal", "out_frac") if k in kw}
    arcs, fin = synth.random_wfst(S, nDist, seed=seed, nWords=50, **gkw)
    inlex, outlex = LexiconPtr("in"), LexiconPtr("out")
    inlex.index("eps", True)
    for i in range(1, nDist + 1):
        inlex.index("</s>" if i == 5 else "d%d" % i, True)                             # one label carries the end-of-sentence name: its runs are not printed (:1096)
    outlex.index("eps", True)
    for i in range(1, 60):
        outlex.index("w%d" % i, True)
    outlex.index("</s>", True)
    wf = WFSTFlyWeightPtr(None, inlex, outlex)
    go = oracle.Wfst()
    for a in arcs:
        go.add_arc(*a); wf._g.add_arc(*a)
    for s, c in fin:
        go.add_final(s, c); wf._g.add_final(s, c)
    dkw = {k: kw[k] for k in ("lmPenalty", "silPenalty") if k in kw}
    silSym = "d%d" % kw["silSym"] if "silSym" in kw else None
    okw = dict(dkw); 
    if silSym:
        okw["silenceX"] = kw["silSym"]
    rng = np.random.default_rng(300 + seed)
    sc = rng.uniform(0, 10, (2, T, nDist)).astype(np.float32)
    nfr = [T, T - 9]
    dec = dsr.Decoder(beam=beam, lmScale=12.0, maxActive=8192, streams=2, latticeTokens=400000, **dkw)
    dsr.check(dsr.load().dsr_decoder_set_symbols(dec.h, wf._g.h, silSym.encode() if silSym else None, b"</s>"))
    out = dec.decode_batch(torch.from_numpy(sc).to(cuda), torch.tensor(nfr, dtype=torch.int32, device=cuda))
    for u in range(2):
        ro = go.decode(sc[u, :nfr[u]], beam=beam, lmScale=12.0, gmmRows=True, **okw)
        assert ro["rc"] == 0 and out[u]["status"] == 0 and ro["score"] == out[u]["score"]
        rows = ro["gmmRows"]; assert rows is not None and len(rows) > 0
        fn = str(tmp_path / ("u%d.gmm" % u))
        open(fn, "w").write("kept\n")                                                  # the reference opens the file for appending (:1023)
        dec.writeGMM(u, "conv7", "A", "spk", "utt%d" % u, 1.25, ro["score"], fn, 0.01)
        exp = "kept\n" + "# %s %10.4f %10.4f\n" % ("utt%d" % u, 1.25, ro["score"])
        skipped = 0
        for inX, startX, endX, score in reversed(rows):
            label = inlex.symbol(inX)
            if label == "</s>":
                skipped += 1; continue
            exp += "%s %s %7.2f %7.2f %-20s %7.2f\n" % ("conv7", "A", 1.25 + startX * 0.01, (endX - startX + 1) * 0.01, label, score)
        assert open(fn).read() == exp
        # the runs tile the utterance: last frame of a run = first frame of the next run - 1 (label runs of the emitting tokens)
        fr = [(s, e) for _, s, e, _ in reversed(rows)]
        assert fr[0][0] == 0 and all(a[1] + 1 == b[0] for a, b in zip(fr, fr[1:])) and fr[-1][1] == nfr[u] - 1
    plain = dsr.Decoder(beam=beam, lmScale=12.0, maxActive=8192, streams=2, **dkw)
    dsr.check(dsr.load().dsr_decoder_set_symbols(plain.h, wf._g.h, None, b"</s>"))
    plain.decode_batch(torch.from_numpy(sc).to(cuda), torch.tensor(nfr, dtype=torch.int32, device=cuda))
    with pytest.raises(dsr.DsrError) as e:
        plain.writeGMM(0, "c", "A", "s", "u", 0.0, 0.0, str(tmp_path / "x.gmm"))
    assert e.value.status == 4                                                         # no bookkeeping: "Must enable lattice generation during decoding."
