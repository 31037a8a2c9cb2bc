"""GPU tests of the stream/feature-operator API, written the way the reference's own driver scripts use it
(asr/test/featureStreamTest.py, btk/tools/filterbank/testNyquistFilterBankDesign.py,
btk/src/superdirectiveBeamformer.cc, asr/test/decodeTest.py) and checked against the oracle."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


class _Iter:
    """The fake audio source of asr/test/featureStreamTest.py:8-35: a Python iterable with size()/reset()."""

    def __init__(self, blocks):
        self._b = blocks

    def size(self):
        return self._b.shape[1]

    def reset(self):
        pass

    def __iter__(self):
        return iter(self._b)


def test_feature_stream_chain_like_featureStreamTest(dsr, oracle, cuda, headset):
    from dsr.btk.stream import PyVectorShortFeatureStreamPtr
    from dsr.btk.feature import (HammingFeaturePtr, FFTFeaturePtr, PowerFeaturePtr, MelFeaturePtr, LogFeaturePtr,
                                 CepstralFeaturePtr, AdjacentFeaturePtr)
    blocks = np.stack([headset[6000 + 160 * t:6000 + 160 * t + 320] for t in range(60)]).astype(np.int16)
    src = PyVectorShortFeatureStreamPtr(_Iter(blocks))
    ham = HammingFeaturePtr(src)
    fft = FFTFeaturePtr(ham, fftLen=512)
    pw = PowerFeaturePtr(fft, powN=257)
    mel = MelFeaturePtr(pw, powN=257, filterN=30)
    lg = LogFeaturePtr(mel)
    cep = CepstralFeaturePtr(lg, ncep=13)
    adj = AdjacentFeaturePtr(cep, delta=5)
    assert (ham.size(), fft.size(), pw.size(), mel.size(), lg.size(), cep.size(), adj.size()) == (320, 512, 257, 30, 30, 13, 143)
    assert cep.name() == "Cepstral" and adj.frameX() == -1
    got = np.stack([np.array(v) for v in adj])                  # __iter__ = reset + next until StopIteration
    assert adj.isEnd()
    # oracle, operator by operator
    o_ham = (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(320) / 319.0)) * blocks.astype(np.float64)
    hamf = o_ham.astype(np.float32)
    spec = np.fft.fft(hamf.astype(np.float64), 512, axis=1)
    o_pw = (spec.real ** 2 + spec.imag ** 2)[:, :257]
    rows = oracle.melbank(257, 16000.0, 0.0, 0.0, 30, 1)
    o_mel = np.stack([np.array([np.dot(o_pw[t, o:o + len(c)], c.astype(np.float64)) for o, c in rows]) for t in range(60)])
    v = o_mel + 1.0
    o_log = np.log10(np.where(v <= 0.0, 1.0, v)).astype(np.float32)    # LogFeature: x+a <= 0 -> 1 (feature.cc:2418-2421)
    o_cep = oracle.sgemv_rows(oracle.cosine_matrix(13, 30, 1), o_log)
    ref = oracle.adjacent(o_cep, 5)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-4
    # individual operator outputs and types
    ham.reset(); v = ham.next(); assert v.dtype == np.float32 and np.array_equal(v, hamf[0])
    f0 = fft.next(); assert f0.dtype == np.complex128 and np.abs(f0 - spec[0]).max() / np.abs(spec[0]).max() < 1e-12
    assert np.abs(f0[512 - 7] - np.conj(f0[7])) == 0.0          # halfComplexUnpack mirror
    p0 = pw.next(); assert p0.dtype == np.float64 and np.abs(p0 - o_pw[0]).max() / o_pw[0].max() < 1e-12
    # protocol: same frame returns the cached vector, skipping a frame is a jindex_error, current() needs a frame
    assert np.array_equal(pw.next(0), p0)
    with pytest.raises(dsr.DsrError) as e:
        pw.next(5)
    assert e.value.status == 6                                   # JINDEX
    lg.reset()
    with pytest.raises(dsr.DsrError):
        lg.current()                                             # "Frame index (-1) < 0." (stream.h:44-46)


def test_full_mfcc_operator_chain_matches_fused_kernel(dsr, oracle, cuda, headset):
    from dsr.btk.feature import (SampleFeaturePtr, PreemphasisFeaturePtr, HammingFeaturePtr, FFTFeaturePtr,
                                 SpectralPowerFeaturePtr, VTLNFeaturePtr, MelFeaturePtr, LogFeaturePtr, CepstralFeaturePtr,
                                 StorageFeaturePtr, MeanSubtractionFeaturePtr, AdjacentFeaturePtr, LinearTransformFeaturePtr)
    import torch
    x = headset[:40000]
    lda = (np.random.default_rng(1234).standard_normal((39, 195)) / np.sqrt(195)).astype(np.float32)
    s = SampleFeaturePtr(blockLen=320, shiftLen=160); s.setSamples(x, 16000)
    chain = LinearTransformFeaturePtr(AdjacentFeaturePtr(MeanSubtractionFeaturePtr(StorageFeaturePtr(CepstralFeaturePtr(LogFeaturePtr(
        MelFeaturePtr(VTLNFeaturePtr(SpectralPowerFeaturePtr(FFTFeaturePtr(HammingFeaturePtr(PreemphasisFeaturePtr(s, mu=0.95)), fftLen=512),
                                                             powN=257), coeffN=257, ratio=1.0, edge=1.0, version=1),
                      powN=257, filterN=30)), ncep=13))), delta=7), sz=39)
    chain.setMatrix(lda)
    got = np.stack([np.array(v) for v in chain])
    ref = oracle.mfcc_chain(x, oracle.mfcc_cfg(lda=lda))
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-4
    fused = dsr.Mfcc(lda=lda).run(torch.from_numpy(x[None]).to(cuda)).cpu().numpy()[0]
    assert np.abs(got - fused[:got.shape[0]]).max() < 1e-5


def test_filterbank_roundtrip_like_testNyquistFilterBankDesign(dsr, oracle, cuda, headset, protos):
    from dsr.btk.feature import SampleFeaturePtr
    from dsr.btk.modulated import OverSampledDFTAnalysisBankPtr, OverSampledDFTSynthesisBankPtr
    from dsr.btk.stream import PyVectorComplexFeatureStreamPtr
    M, m, r, h, g = protos["M512-m2-r2"]
    D = M >> r
    x = headset[:20000]
    sampleFeature = SampleFeaturePtr(blockLen=D, shiftLen=D, padZeros=True)
    analysisFB = OverSampledDFTAnalysisBankPtr(sampleFeature, prototype=h, M=M, m=m, r=r, delayCompensationType=2)
    synthesisFB = OverSampledDFTSynthesisBankPtr(PyVectorComplexFeatureStreamPtr(analysisFB), prototype=g, M=M, m=m, r=r, delayCompensationType=2)
    sampleFeature.setSamples(x, 16000)
    wavebuffer = []
    for b in synthesisFB:
        wavebuffer.extend(np.array(b))
    y = np.array(wavebuffer) * float(D)
    n = min(len(x), len(y))
    assert np.sqrt(np.mean((y[2000:n - 2000] - x[2000:n - 2000]) ** 2)) / np.sqrt(np.mean(x ** 2)) < 2e-5
    # the analysis stream itself: M complex doubles per frame, full Hermitian vector
    analysisFB.reset(); X0 = np.array(analysisFB.next())
    ref = oracle.analysis_bank(x, h, M, m, r, 2)
    assert X0.shape == (M,) and np.abs(X0 - ref[0]).max() / (np.abs(ref[0]).max() + 1e-9) < 2e-5
    with pytest.raises(dsr.DsrError) as e:
        OverSampledDFTAnalysisBankPtr(SampleFeaturePtr(blockLen=100, shiftLen=100, padZeros=True), prototype=h, M=M, m=m, r=r)
    assert e.value.status == 5                                   # jdimension_error "Input block length != _D"


def test_synthesis_of_frames_that_are_not_conjugate_symmetric(dsr, oracle, cuda, protos):
    """OverSampledDFTSynthesisBank transforms all M bins of a frame and keeps the real part (modulated.cc:598-610): SubbandMMI with the APAB post-filter
    hands over frames that are not conjugate-symmetric, and the operator has to synthesise their Hermitian part, not their lower half."""
    from dsr.btk.modulated import OverSampledDFTSynthesisBankPtr
    from dsr.btk.stream import PyVectorComplexFeatureStreamPtr
    M, m, r, h, g = protos["M256-m4-r1"]
    T = 40
    rng = np.random.default_rng(77)
    rows = rng.standard_normal((T, M)) + 1j * rng.standard_normal((T, M))

    class Frames:
        def size(self): return M
        def reset(self): pass
        def __iter__(self): return iter(rows)
    out = np.concatenate([np.array(b) for b in OverSampledDFTSynthesisBankPtr(PyVectorComplexFeatureStreamPtr(Frames()), prototype=g, M=M, m=m, r=r)])
    ref = oracle.synthesis_bank(rows, g, M, m, r, 0)
    lower = rows.copy(); lower[:, M // 2 + 1:] = np.conj(rows[:, 1:M // 2][:, ::-1])
    assert np.abs(oracle.synthesis_bank(lower, g, M, m, r, 0) - ref).max() > 0.1 * np.abs(ref).max()      # the two readings differ by far
    assert out.shape == ref.shape and np.abs(out - ref).max() / np.sqrt(np.mean(ref ** 2)) < 5e-5


def test_mvdr_driver_like_superdirectiveBeamformer(dsr, oracle, cuda, protos):
    from dsr.btk.feature import SampleFeaturePtr
    from dsr.btk.modulated import OverSampledDFTAnalysisBankPtr, OverSampledDFTSynthesisBankPtr
    from dsr.btk.beamformer import SubbandMVDRPtr, calcDelaysPolar2
    M, m, r, h, g = protos["M256-m4-r1"]
    D, Cn, n = M >> r, 8, 12000
    x = synth.array_signal(n, Cn, seed=3)
    mp = synth.linear_array(Cn)
    beamformer = SubbandMVDRPtr(fftLen=M, halfBandShift=False)
    for c in range(Cn):
        s = SampleFeaturePtr(blockLen=D, shiftLen=D, padZeros=True); s.setSamples(x[c], 16000)
        beamformer.setChannel(OverSampledDFTAnalysisBankPtr(s, prototype=h, M=M, m=m, r=r))
    delays = calcDelaysPolar2(np.deg2rad(30.0), np.pi / 2, mp)
    beamformer.calcArrayManifoldVectors(16000.0, delays)
    beamformer.setDiffuseNoiseModel(mp, 16000.0, 343740.0)
    beamformer.divideAllNonDiagonalElements(0.01)
    beamformer.calcMVDRWeights(16000.0, 1.0E-8)
    synthesisFB = OverSampledDFTSynthesisBankPtr(beamformer, prototype=g, M=M, m=m, r=r)
    out = np.concatenate([np.array(b) for b in synthesisFB])
    W = beamformer._weights().get(4)
    Xc = np.stack([oracle.analysis_bank(x[c], h, M, m, r, 0) for c in range(Cn)])
    ref = oracle.synthesis_bank(oracle.beamform_apply(Xc, W), g, M, m, r, 0)
    assert out.shape == ref.shape and np.abs(out - ref).max() / np.sqrt(np.mean(ref ** 2)) < 5e-5


def test_decode_like_decodeTest(dsr, oracle, cuda, headset, tmp_path):
    """asr/test/decodeTest.py shape: description files + binary model files + lexica + WFST file -> decode() / bestHypo()."""
    from dsr.btk.feature import (SampleFeaturePtr, HammingFeaturePtr, FFTFeaturePtr, SpectralPowerFeaturePtr, MelFeaturePtr,
                                 LogFeaturePtr, CepstralFeaturePtr, FeatureSetPtr)
    from dsr.asr.dictionary import LexiconPtr
    from dsr.asr.gaussian import CodebookSetBasicPtr, DistribSetBasicPtr
    from dsr.asr.decoder import WFSTFlyWeightPtr, DecoderFlyWeightPtr
    K, R, D = 24, 8, 13
    m = synth.gmm_model(K, R, D, seed=5); m["mean"] *= 20.0
    cbf, dsf = str(tmp_path / "cb.bin"), str(tmp_path / "ds.bin")
    dsr.Gmm(**m).save(cbf, dsf)
    open(tmp_path / "cb.desc", "w").write("; codebooks\n" + "".join("%-25s%-20s%-10d%-3d%-10s\n" % ("cb%d" % k, "Cepstral", R, D, "DIAGONAL") for k in range(K)))
    open(tmp_path / "ds.desc", "w").write("".join("%-25s%-25s\n" % ("ds%d" % k, "cb%d" % k) for k in range(K)))
    inlex = ["eps"] + ["ds%d" % k for k in range(K)]; inlex[4] = "SIL-m"
    outlex = ["eps", "</s>"] + ["w%d" % i for i in range(50)]
    open(tmp_path / "in.lex", "w").write("\n".join(inlex) + "\n"); open(tmp_path / "out.lex", "w").write("\n".join(outlex) + "\n")
    arcs, fin = synth.random_wfst(400, K, seed=4, nWords=51, out_frac=0.2)
    go = oracle.Wfst()
    with open(tmp_path / "g.fsm", "w") as f:
        for a in arcs:
            f.write("%d %d %d %d %.9g\n" % a); go.add_arc(*a)
        for s, c in fin:
            f.write("%d %.9g\n" % (s, c)); go.add_final(s, c)
    samp = SampleFeaturePtr(blockLen=320, shiftLen=160)
    feat = CepstralFeaturePtr(LogFeaturePtr(MelFeaturePtr(SpectralPowerFeaturePtr(FFTFeaturePtr(HammingFeaturePtr(samp), fftLen=512), powN=257),
                                                         powN=257, filterN=30)), ncep=13)
    fs = FeatureSetPtr(); fs.add(feat)
    cbs = CodebookSetBasicPtr(str(tmp_path / "cb.desc"), fs, cbf)
    dss = DistribSetBasicPtr(cbs, str(tmp_path / "ds.desc"), dsf)
    wfst = WFSTFlyWeightPtr(LexiconPtr("state"), LexiconPtr("in", str(tmp_path / "in.lex")), LexiconPtr("out", str(tmp_path / "out.lex")))
    wfst.read(str(tmp_path / "g.fsm"), binary=False)
    d = DecoderFlyWeightPtr(dss, beam=80.0, lmScale=12.0, silPenalty=0.5)
    d.set(wfst)
    samp.setSamples(headset[20000:36000], 16000)
    score = d.decode(); hyp = d.bestHypo()
    # oracle on the device features
    X = np.stack([np.array(v) for v in feat])
    cb = oracle.Codebooks(m["refN"], m["mean"], m["ivar"], m["det"])
    so, _ = oracle.gmm_score_opt(cb, m["val"], X)
    ro = go.decode(so, beam=80.0, lmScale=12.0, silPenalty=0.5, silenceX=4)
    assert score == ro["score"] and np.array_equal(d.bestArcs(), ro["arcs"])
    assert hyp == "".join(outlex[w] + " " for w in ro["words"])
    bad = DecoderFlyWeightPtr(dss, silSymbol="NOPE")
    with pytest.raises(dsr.DsrError) as e:
        bad.set(wfst)
    assert e.value.status == 11                                  # jkey_error
    # ---- the rest of the ASR boundary: bestPath, finalStatesN, bestHypo(useInputSymbols), traceBackSucceeded
    ex = go.export(); ain = ex["arcIn"][ro["arcs"]]
    assert list(d.bestPath()) == [inlex[i] for i in ain if i != 0]                                   # decoder.h:775-797: names along the path
    kept, lastX = [], 0
    for i in ain[::-1]:                                                                             # decoder.h:754-760, walking prev() from the end
        if i != 0 and i != lastX:
            kept.append(int(i)); lastX = int(i)
    assert d.bestHypo(useInputSymbols=True) == "".join(inlex[i] + " " for i in kept[::-1])
    assert d.finalStatesN() == ro["finalStatesN"] and d.traceBackSucceeded() == ro["reachedFinal"]
    # ---- Distrib::score(frameX) / DistribSet::find (distribBasic.h:48-50,183-190) through the feature-stream protocol
    assert dss.ndists() == K and dss.index("ds7") == 7 and dss.find("ds7").name() == "ds7"
    with pytest.raises(dsr.DsrError) as e:
        dss.find("nope")
    assert e.value.status == 11
    dss.resetFeature(); dss.resetCache()
    for t in (0, 1, 2):
        for k in (0, 7, K - 1):
            assert dss.find(k).score(t) == so[t, k]                                                  # the same bits the decoder consumed
    with pytest.raises(dsr.DsrError) as e:
        dss.find(0).score(9)                                                                         # out of order: jindex_error (feature.cc:1222-1223)
    assert e.value.status == 6
    # ---- lattice() behind the same face, against the oracle
    ro2 = go.decode(so, beam=80.0, lmScale=12.0, silPenalty=0.5, silenceX=4, lattice=True, eosX=1)
    Ld = d.lattice().data
    for k in ("nodeFinal", "from", "to", "in", "out", "start", "end"):
        assert np.array_equal(Ld[k], ro2["lattice"][k]), k
    assert np.array_equal(Ld["ac"].view(np.int64), ro2["lattice"]["ac"].view(np.int64)) and np.array_equal(Ld["lm"].view(np.int64), ro2["lattice"]["lm"].view(np.int64))
    # ---- asr.lattice.LatticePtr on that lattice (lattice.i:79-123): rescore with the search's own parameters finds the search's hypothesis again;
    #      gammaProbsDist recomputes every link's acoustic score from the distribution set (frames scored on the device, one gather kernel for the sums)
    import importlib, os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    olat = importlib.import_module("oracle_lattice")
    lat = d.lattice(); O = olat.Lattice.from_arrays(ro2["lattice"])
    s1 = lat.rescore(12.0, 0.0, 0.5, "SIL-m"); o1 = O.rescore(12.0, 0.0, 0.5, 4)
    assert np.float32(s1).view(np.int32) == np.float32(o1).view(np.int32)
    assert lat.bestHypo() == hyp and lat.bestHypo(True) == "".join(inlex[i] + " " for i in O.bestHypo(True))
    for ed in O.allEdges:                                                                            # _updateAcNode (lattice.cc:392-409) on the oracle's scores
        if ed.input != 0 and not ed.prev.final:
            acc = 0.0
            for t in range(ed.start, ed.end + 1):
                acc += float(so[t, ed.input - 1])
            ed.ac = acc
    g1 = lat.gammaProbsDist(dss, 1.0 / 12.0, 12.0, 0.0, 0.5, "SIL-m"); O._clearSorted(); g0 = O.gammaProbs(1.0 / 12.0, 12.0, 0.0, 0.5, 4)
    assert g1 == g0
    sa, sb = lat.state(), O.state()
    for k in ("gamma", "fwd", "bwd"):
        assert np.array_equal(sa[k].view(np.int64), sb[k].view(np.int64)), k
    lat.write(str(tmp_path / "p.lat"), writeData=True); O.write(str(tmp_path / "o.lat"), writeData=True)
    assert open(tmp_path / "p.lat", "rb").read() == open(tmp_path / "o.lat", "rb").read()          # the recomputed ac column and the posteriors, byte for byte
    lat.writeCTM("conv", "1", "spk", "utt", 0.0, float(score), str(tmp_path / "p.ctm")); O.writeCTM(outlex, "conv", "1", "spk", "utt", 0.0, float(score), str(tmp_path / "o.ctm"))
    assert open(tmp_path / "p.ctm", "rb").read() == open(tmp_path / "o.ctm", "rb").read() and " w" in open(tmp_path / "p.ctm").read()
    with pytest.raises(dsr.DsrError) as e:
        lat.rescore(12.0, 0.0, 0.0, "NOPE")
    assert e.value.status == 11                                                                      # inputLexicon()->index(silSymbol): jkey_error
    # ---- topN (decoder.h:571-581): the 6 best tokens of every list expanded in score order, no beam
    dn = DecoderFlyWeightPtr(dss, beam=80.0, lmScale=12.0, silPenalty=0.5, topN=6, generateLattice=False); dn.set(wfst)
    sn = dn.decode()
    rn = go.decode(so, beam=80.0, lmScale=12.0, silPenalty=0.5, silenceX=4, topN=6)
    assert rn["rc"] == 0 and sn == rn["score"] and np.array_equal(dn.bestArcs(), rn["arcs"]) and sn != score
    # ---- the same transducer written with symbols instead of numbers (wfstFlyWeight.cc:311-347) reads to the same graph
    with open(tmp_path / "gsym.fsm", "w") as f:
        for a in arcs:
            f.write("%d %d %s %s %.9g\n" % (a[0], a[1], inlex[a[2]], outlex[a[3]], a[4]))
        for s_, c in fin:
            f.write("%d %.9g\n" % (s_, c))
    w2 = WFSTFlyWeightPtr(LexiconPtr("state"), LexiconPtr("in", str(tmp_path / "in.lex")), LexiconPtr("out", str(tmp_path / "out.lex")))
    w2.read(str(tmp_path / "gsym.fsm"), binary=False)
    e1, e2 = wfst._g.export(), w2._g.export()
    for k in e1:
        assert np.array_equal(e1[k], e2[k]), k
    assert w2.hasFinalState() and w2.inputLexicon().symbol(4) == "SIL-m"
    # an empty utterance: the exception escapes decode() (decoder.h:691)
    samp.setSamples(np.zeros(10, np.float32), 16000)
    with pytest.raises(StopIteration):
        d.decode()


def test_python_source_is_refilled_when_a_downstream_operator_resets(dsr, cuda, headset):
    """PyFeatureStream::reset() calls the Python object's reset() and iterates it afresh (pyStream.h:100-130).  The usual driver
    resets or iterates the LAST operator of a chain; that reset cascades to the source inside the library, which must then pull the
    Python iterable again -- a chain that kept serving the previous utterance's frames would return stale output silently."""
    from dsr.btk.stream import PyVectorShortFeatureStreamPtr
    from dsr.btk.feature import HammingFeaturePtr, FFTFeaturePtr

    class Utterances:
        """one utterance per reset(), as a Python driver feeding a list of files would"""

        def __init__(self, utts):
            self.utts, self.k, self.resets = utts, -1, 0

        def size(self):
            return 320

        def reset(self):
            self.resets += 1; self.k = min(self.k + 1, len(self.utts) - 1)

        def __iter__(self):
            return iter(self.utts[max(self.k, 0)])

    mk = lambda off, n: np.stack([headset[off + 160 * t:off + 160 * t + 320] for t in range(n)]).astype(np.int16)
    u0, u1 = mk(6000, 12), mk(30000, 7)
    it = Utterances([u0, u1])
    src = PyVectorShortFeatureStreamPtr(it)
    fft = FFTFeaturePtr(HammingFeaturePtr(src), fftLen=512)
    win = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(320) / 319.0)
    ref = lambda b: np.fft.fft((win * b.astype(np.float64)).astype(np.float32).astype(np.float64), 512, axis=1)
    a = np.stack([np.array(v) for v in fft])                    # __iter__ resets the LAST operator only
    assert a.shape[0] == 12 and np.abs(a - ref(u0)).max() <= 1e-9 * np.abs(ref(u0)).max()
    r0 = it.resets
    b = np.stack([np.array(v) for v in fft])                    # second utterance: the source must have been reset and drained again
    assert it.resets == r0 + 1
    assert b.shape[0] == 7 and np.abs(b - ref(u1)).max() <= 1e-9 * np.abs(ref(u1)).max()

    class Broken(Utterances):
        def reset(self):
            Utterances.reset(self)
            if self.resets >= 1:
                raise RuntimeError("reset failed in Python")

    src2 = PyVectorShortFeatureStreamPtr(Broken([u0, u1]))
    ham2 = HammingFeaturePtr(src2)
    ham2.reset()
    with pytest.raises((RuntimeError, dsr.DsrError)):           # JPYTHON surfaces (jexception.i:181-183), never stale frames
        ham2.next()


def test_legacy_formats_behind_the_operators(dsr, oracle, cuda, headset, tmp_path):
    """StorageFeature::write/read (feature.cc:3025-3067, quirks kept), LinearTransformFeature::load with GSL and Janus FMAT
    files (feature.cc:2972-2976, gslmatrix.cc:27-96), the older Janus codebook-set format (codebookBasic.cc:311-350,934-957)."""
    import struct
    import torch
    from dsr.btk.feature import (SampleFeaturePtr, HammingFeaturePtr, FFTFeaturePtr, SpectralPowerFeaturePtr, MelFeaturePtr, LogFeaturePtr,
                                 CepstralFeaturePtr, StorageFeaturePtr, LinearTransformFeaturePtr)
    samp = SampleFeaturePtr(blockLen=320, shiftLen=160)
    cep = CepstralFeaturePtr(LogFeaturePtr(MelFeaturePtr(SpectralPowerFeaturePtr(FFTFeaturePtr(HammingFeaturePtr(samp), fftLen=512), powN=257),
                                                        powN=257, filterN=30)), ncep=13)
    st = StorageFeaturePtr(cep)
    samp.setSamples(headset[9000:9000 + 160 * 30], 16000)
    last = st.evaluate(); assert last == st.frameX() and last >= 20
    rows = np.stack([np.array(st.next(t)) for t in range(last + 1)])
    # binary: big-endian _frameX (the LAST INDEX, not the count) and size, then the frames 0.._frameX as native float blocks
    st.write(str(tmp_path / "s.bin"))
    assert open(tmp_path / "s.bin", "rb").read() == struct.pack(">ii", last, 13) + rows.tobytes()
    st.write(str(tmp_path / "s.txt"), plainText=True)
    txt = open(tmp_path / "s.txt").read().splitlines()
    assert txt[0] == "%d %d" % (last, 13) and len(txt) == last + 2 and txt[1] == " ".join("%g" % v for v in rows[0])
    # read(): _frameX = the number in the file; that many frames are read (one fewer than were written); the rest of the store is as it was
    st2 = StorageFeaturePtr(CepstralFeaturePtr(LogFeaturePtr(MelFeaturePtr(SpectralPowerFeaturePtr(FFTFeaturePtr(HammingFeaturePtr(SampleFeaturePtr(blockLen=320, shiftLen=160)), fftLen=512),
                                                                                              powN=257), powN=257, filterN=30)), ncep=13))
    st2.read(str(tmp_path / "s.bin"))
    assert st2.frameX() == last
    got = np.stack([np.array(st2.next(t)) for t in range(last + 1)])
    assert np.array_equal(got[:last], rows[:last]) and np.all(got[last] == 0)
    open(tmp_path / "bad.bin", "wb").write(struct.pack(">ii", 3, 12) + b"\0" * 200)
    with pytest.raises(dsr.DsrError) as e:
        st2.read(str(tmp_path / "bad.bin"))
    assert e.value.status == 5                                            # "Feature dimensions (12 vs. 13) do not match."
    # LinearTransformFeature::load
    lt = LinearTransformFeaturePtr(st, 5)
    A = np.random.default_rng(3).standard_normal((5, 13)).astype(np.float32)
    open(tmp_path / "a.gsl", "wb").write(A.tobytes())
    open(tmp_path / "a.fmat", "wb").write(b"FMAT" + struct.pack(">ii", -1, 13) + struct.pack(">f", 0.0) + A.astype(">f4").tobytes())
    for fn, old in (("a.gsl", False), ("a.fmat", True)):
        lt.load(str(tmp_path / fn), old); lt.reset()
        y = np.stack([np.array(v) for v in lt])
        assert np.abs(y - oracle.sgemv_rows(A, rows)).max() < 1e-4
    open(tmp_path / "small.fmat", "wb").write(b"FMAT" + struct.pack(">ii", 4, 13) + struct.pack(">f", 0.0) + A[:4].astype(">f4").tobytes())
    with pytest.raises(dsr.DsrError) as e:
        lt.load(str(tmp_path / "small.fmat"), True)                       # the crop back to 5 x 13 cannot grow the matrix (gslmatrix.cc:6-15)
    assert e.value.status == 5
    # the older Janus codebook-set format: written by save(janusFormat = true), read when the file does not start with CodebookMagic
    import ctypes as C
    m = synth.gmm_model(6, 4, 13, seed=9); g = dsr.Gmm(**m)
    cbj, dsf, cbn = str(tmp_path / "cb.janus"), str(tmp_path / "ds.bin"), str(tmp_path / "cb.new")
    dsr.check(dsr.load().dsr_gmm_save_janus(g.h, cbj.encode(), dsf.encode())); g.save(cbn, dsf)
    be = lambda a: np.asarray(a, np.float32).astype(">f4").tobytes()
    exp = struct.pack(">i", 6)
    for k in range(6):
        nm = ("cb%d" % k).encode()
        exp += struct.pack(">h", len(nm)) + nm + b"\0" + struct.pack(">iii", 4, 13, 2)
        for j in range(4 * k, 4 * k + 4):
            exp += be(m["mean"][j]) + be(m["ivar"][j]) + be(m["det"][j])
    assert open(cbj, "rb").read() == exp
    g2 = dsr.Gmm(files=(cbj, dsf)); g3 = dsr.Gmm(files=(cbn, dsf))
    x = torch.from_numpy(np.random.default_rng(4).standard_normal((50, 13)).astype(np.float32)).to(cuda)
    assert torch.equal(g2.score(x)[0], g3.score(x)[0]) and torch.equal(g2.score(x)[0], g.score(x)[0])
    # per-Gaussian counts and types (uniform type -1), and the reference's refusals
    var = struct.pack(">i", 1) + struct.pack(">h", 3) + b"cb0\0" + struct.pack(">iii", 4, 13, -1)
    for j in range(4):
        var += be([7.0]) + be(m["mean"][j]) + struct.pack(">i", 2) + be(m["ivar"][j]) + be(m["det"][j])
    open(tmp_path / "cb.var", "wb").write(var)
    one = dsr.Gmm(refN=[4], mean=m["mean"][:4], ivar=m["ivar"][:4], det=m["det"][:4], val=m["val"][:4]); one.save(str(tmp_path / "x.cb"), str(tmp_path / "one.ds"))
    g4 = dsr.Gmm(files=(str(tmp_path / "cb.var"), str(tmp_path / "one.ds")))
    assert torch.equal(g4.score(x)[0], one.score(x)[0])
    open(tmp_path / "cb.full", "wb").write(var.replace(struct.pack(">i", 2), struct.pack(">i", 3), 1))      # a full covariance: "Wrong covariance type."
    open(tmp_path / "cb.comp", "wb").write(struct.pack(">i", -1) + var[4:])                                 # compressed mode: "Mode not supported."
    for fn in ("cb.full", "cb.comp"):
        with pytest.raises(dsr.DsrError) as e:
            dsr.Gmm(files=(str(tmp_path / fn), str(tmp_path / "one.ds")))
        assert e.value.status == 8


@pytest.mark.parametrize("runon,dnf", [(False, 0.0), (False, 3.0), (True, 0.0), (True, 2.0)])
def test_mean_subtraction_with_frame_weights(dsr, oracle, cuda, runon, dnf):
    """MeanSubtractionFeature(src, weight, devNormFactor, runon) (feature.cc:2577-2707): element 0 of the weight stream's frames weighs the batch
    statistics; in run-on mode frames with weight <= 0 are normalised without updating them"""
    from dsr.btk.feature import MeanSubtractionFeaturePtr
    from dsr.btk.stream import PyVectorFloatFeatureStreamPtr
    T, N = 700, 13
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((T, N)) * 3.0 + 1.5).astype(np.float32)
    w = np.zeros((T, 2), np.float32); w[:, 0] = (rng.random(T) > 0.3) * rng.uniform(0.2, 1.0, T); w[:, 1] = 7.0      # only element 0 counts

    class Frames:
        def __init__(self, a): self.a = a
        def size(self): return self.a.shape[1]
        def reset(self): pass
        def __iter__(self): return iter(self.a)
    src = PyVectorFloatFeatureStreamPtr(Frames(x)); wsrc = PyVectorFloatFeatureStreamPtr(Frames(w))
    op = MeanSubtractionFeaturePtr(src, weight=wsrc, devNormFactor=dnf, runon=runon)
    got = np.stack([np.array(v) for v in op])
    ref = oracle.cmn_runon(x, dnf, weights=w[:, 0]) if runon else oracle.cmn_batch(x, dnf, weights=w[:, 0])[0]
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-5
    plain = oracle.cmn_runon(x, dnf) if runon else oracle.cmn_batch(x, dnf)[0]
    assert np.abs(ref - plain).max() > 1e-3                                  # the weights matter


def test_sample_feature_read_formats_and_branches(dsr, cuda, tmp_path):
    """SampleFeature::read (feature.cc:243-393): channel selection, sample range, the integer scale of norm == 0 and libsndfile's float normalisation
    otherwise, and the reference's error branches -- on PCM WAV files of 8, 16, 24 and 32 bits"""
    import wave
    from dsr.btk.feature import SampleFeaturePtr
    rng = np.random.default_rng(9)
    n, nch = 1000, 2
    for sw in (1, 2, 3, 4):
        bits = 8 * sw
        v = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(n, nch), dtype=np.int64)
        fn = str(tmp_path / ("s%d.wav" % bits))
        with wave.open(fn, "wb") as w:
            w.setnchannels(nch); w.setsampwidth(sw); w.setframerate(16000)
            if sw == 1:
                raw = (v + 128).astype(np.uint8).tobytes()
            elif sw == 3:
                u = (v & 0xFFFFFF).astype(np.uint32)
                raw = np.stack([u & 255, (u >> 8) & 255, (u >> 16) & 255], axis=-1).astype(np.uint8).tobytes()
            else:
                raw = v.astype("<i%d" % sw).tobytes()
            w.writeframes(raw)
        s = SampleFeaturePtr(blockLen=100, shiftLen=100)
        assert s.read(fn, chX=2) == n and s.getSampleRate() == 16000
        got = np.concatenate([np.array(b) for b in s])
        assert len(got) >= n - 100 and np.array_equal(got, v[:len(got), 1].astype(np.float32))  # norm == 0: the file's integer scale (the framing may hold back the last block)
        assert s.read(fn, chX=1, cfrom=100, to=499, norm=1.0) == 400
        got = np.concatenate([np.array(b) for b in s])
        assert len(got) >= 300 and np.array_equal(got, (v[100:100 + len(got), 0] / float(1 << (bits - 1))).astype(np.float32))
        s.read(fn, chX=1, cfrom=100, to=499, norm=3.0)
        got3 = np.concatenate([np.array(b) for b in s])
        assert np.array_equal(got3, got * np.float32(3.0))
        with pytest.raises(dsr.DsrError) as e:
            s.read(fn, chX=0)
        assert e.value.status == 4 and "Multi-channel" in str(e.value)
        with pytest.raises(dsr.DsrError) as e:
            s.read(fn, chX=3)
        assert e.value.status == 4
        with pytest.raises(IOError):
            s.read(fn, cfrom=900, to=100)
        with pytest.raises(dsr.DsrError):
            s.read(fn, outsamplerate=8000)
    with pytest.raises(IOError):
        SampleFeaturePtr(blockLen=100, shiftLen=100).read(str(tmp_path / "missing.wav"))
