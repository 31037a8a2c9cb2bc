"""DecoderWordTrace (asr/decoder/decoder.h:1146-1304, asr/decoder/decoder.cc:126-470): the device search (generateLattice = false) against the restatement
of the shipped class (oracle/oracle_wordtrace.py) -- best score, its two halves, final-state count, active hypotheses, the one symbol the shipped
bestHypo() returns, and the word sequence along the word traces, all exact; and where both sides stop with generateLattice = true."""
import numpy as np
import pytest

from oracle import oracle_wfst as OW
from oracle import oracle_wordtrace as WT
from tests import synth

pytestmark = pytest.mark.gpu


def _build(S, nDist, seed, eps_frac, out_frac, nFinal, ties):
    arcs, fin = synth.random_wfst(S, nDist, seed=seed, outdeg=3, eps_frac=eps_frac, out_frac=out_frac, nWords=40, nFinal=nFinal, ties=ties)
    return arcs, fin


def _oracle(arcs, fin, scores, **kw):
    g = OW.FlyWeightSortedOutput()
    for a in arcs:
        g.add_arc(*a)
    for s, c in fin:
        g.add_final(s, c)
    T = scores.shape[0]

    def scoreFn(distX, frameX):
        if frameX >= T:
            raise WT.EndOfSamples()
        return float(scores[frameX, distX - 1])
    d = WT.DecoderWordTrace(scoreFn, **kw); d.set(g)
    return d


@pytest.mark.parametrize("seed,S,eps,outf,lmPen,silPen,silX,insSil,ties,beam", [
    (1, 60, 0.10, 0.30, 0.0, 0.0, 0xFFFFFFFF, False, False, 40.0),
    (2, 80, 0.20, 0.50, 0.7, 0.0, 0xFFFFFFFF, False, True, 25.0),
    (3, 50, 0.15, 0.30, 0.3, 1.1, 3, False, True, 30.0),
    (4, 70, 0.25, 0.20, 0.0, 0.9, 2, True, False, 35.0),           # insertSilence: a trace with word 0 on entering silence
    (5, 40, 0.30, 0.60, 1.3, 0.4, 0, True, True, 60.0),            # silenceX == 0: the epsilon arcs count as silence
    (6, 90, 0.05, 0.10, 0.0, 0.0, 0xFFFFFFFF, False, True, 12.0),  # narrow beam
])
def test_wordtrace_search_matches_the_restatement(dsr, cuda, seed, S, eps, outf, lmPen, silPen, silX, insSil, ties, beam):
    import torch
    nDist, T, U = 10, 40, 3
    arcs, fin = _build(S, nDist, seed, eps, outf, 6, ties)
    rng = np.random.default_rng(seed)
    sc = (rng.integers(0, 8, (U, T, nDist)).astype(np.float32) if ties else rng.uniform(0, 8, (U, T, nDist)).astype(np.float32))
    nfr = [T, T - 7, 1]
    g = dsr.Wfst(); dsr.check(dsr.load().dsr_wfst_set_sorted_output(g.h, 1))
    for a in arcs:
        g.add_arc(*a)
    for s, c in fin:
        g.add_final(s, c)
    dec = dsr.Decoder(beam=beam, lmScale=9.5, lmPenalty=lmPen, silPenalty=silPen, silenceX=silX, maxActive=4096, streams=2, wordTrace=1, generateLattice=False,
                      insertSilence=insSil)
    dec.set(g)
    out = dec.decode_batch(torch.from_numpy(sc).to(cuda), torch.tensor(nfr, dtype=torch.int32, device=cuda), maxPath=256)
    ex = g.export()
    for u in range(U):
        o = _oracle(arcs, fin, sc[u, :nfr[u]], beam=beam, lmScale=9.5, lmPenalty=lmPen, silPenalty=silPen, silenceX=silX, generateLattice=False, insertSilence=insSil)
        score = o.decode()
        r = out[u]
        assert r["status"] == 0
        assert r["score"] == score and r["ac"] == o.best.ac and r["lm"] == o.best.lm, (u, r["score"], score)
        assert r["reachedFinal"] == int(o.reachedFinal) and r["finalStatesN"] == o.finalStatesN() and r["activeHypos"] == o.activeHypos
        assert list(r["words"]) == o.wordTraceIds(), u
        a = int(r["arcs"][0])                                               # the best token's own edge: bestHypo's one symbol
        assert int(ex["arcOut"][a]) == o.best.edge.output and int(ex["arcIn"][a]) == o.best.edge.input


def test_wordtrace_generate_lattice_is_where_both_sides_stop(dsr, cuda):
    """generateLattice = true (the reference's default): _placeOnList merges two tokens' worse chains and reads wordTrace()->wordSequenceX() of each
    (decoder.cc:239).  On a graph whose first arcs carry no output symbol the first recombination meets tokens without a word trace: the restatement
    stops there (NullWordTrace, with frame and state); the product refuses the decode.  On a graph where EVERY arc has an output symbol the shipped
    merge runs -- with _notPresent comparing _uniqueIndices[0] only (:203-211) -- and the restatement decodes; the product still refuses (not built)."""
    import torch
    nDist, T = 6, 12
    arcs, fin = _build(30, nDist, 11, 0.0, 0.0, 4, True)                    # no output symbols at all
    sc = np.random.default_rng(2).integers(0, 5, (1, T, nDist)).astype(np.float32)
    o = _oracle(arcs, fin, sc[0], beam=50.0, lmScale=9.5, generateLattice=True)
    with pytest.raises(WT.NullWordTrace) as e:
        o.decode()
    assert "frame" in str(e.value)
    g = dsr.Wfst(); dsr.check(dsr.load().dsr_wfst_set_sorted_output(g.h, 1))
    for a in arcs:
        g.add_arc(*a)
    for s, c in fin:
        g.add_final(s, c)
    dec = dsr.Decoder(beam=50.0, lmScale=9.5, maxActive=1024, streams=1, wordTrace=1, generateLattice=True); dec.set(g)
    with pytest.raises(dsr.DsrError) as e2:
        dec.decode_batch(torch.from_numpy(sc).to(cuda), maxPath=16)
    assert "JCONSISTENCY" in str(e2.value) and "decoder.cc:239" in str(e2.value)
    # every arc with an output symbol: the reference's merge is defined; the restatement runs it as shipped
    arcs2 = [(a[0], a[1], a[2], 1 + (k % 7), a[4]) for k, a in enumerate(arcs)]
    o2 = _oracle(arcs2, fin, sc[0], beam=50.0, lmScale=9.5, generateLattice=True, fastHash=True)
    s2 = o2.decode()
    o3 = _oracle(arcs2, fin, sc[0], beam=50.0, lmScale=9.5, generateLattice=False)
    assert np.isfinite(s2) and s2 >= o3.decode() - 1e-3                     # the merged search cannot beat the plain 1-best by more than float noise


def test_sorted_output_container(dsr):
    """WFSTFlyWeightSortedOutput::Node::_addEdgeForce (wfstFlyWeight.cc:754-776): arcs by (output, input), a new arc in front of its equals"""
    arcs = [(0, 1, 5, 2, 1.0), (0, 2, 3, 2, 2.0), (0, 3, 5, 0, 3.0), (0, 4, 5, 2, 4.0), (0, 5, 1, 7, 5.0), (0, 6, 0, 0, 6.0), (1, 0, 2, 2, 0.5)]
    g = dsr.Wfst(); dsr.check(dsr.load().dsr_wfst_set_sorted_output(g.h, 1))
    o = OW.FlyWeightSortedOutput()
    for a in arcs:
        g.add_arc(*a); o.add_arc(*a)
    ex = g.export()
    n0 = int(ex["arcOff"][1])
    got = [(int(ex["arcOut"][k]), int(ex["arcIn"][k]), float(ex["arcCost"][k])) for k in range(n0)]
    want = [(e.output, e.input, e.cost) for e in o.initial.iter_edges()]
    assert got == want == [(0, 0, 6.0), (0, 5, 3.0), (2, 3, 2.0), (2, 5, 4.0), (2, 5, 1.0), (7, 1, 5.0)]
