"""asr.decoder: WFSTFlyWeightPtr / DecoderFlyWeightPtr (decoder.i:52-70,147-199)."""
import numpy as np

from .. import _capi as K


class WFSTFlyWeightPtr(object):
    def __init__(self, statelex=None, inlex=None, outlex=None, name="WFSTFlyWeight"):
        self._g = K.Wfst(); self._state, self._in, self._out = statelex, inlex, outlex

    def read(self, fileName, binary=False):
        self._g.read(fileName, binary)

    def write(self, fileName, binary=True, useSymbols=False):
        self._g.write(fileName, binary)

    def hasFinalState(self):
        return bool(self._g.export()["nodeFinal"].any())

    def inputLexicon(self):
        return self._in

    def outputLexicon(self):
        return self._out


class WFSTransducerPtr(WFSTFlyWeightPtr):
    """The dynamic container (asr/fsm/fsm.h WFSTransducer): read(fileName, noSelfLoops) per asr/fsm/fsm.cc:901-986.  Node and arc
    order (initial node of its own, arcs prepended, epsilon:epsilon self loops dropped) are those of the fly-weight reader."""

    def __init__(self, statelex=None, inlex=None, outlex=None, name="WFSTransducer"):
        WFSTFlyWeightPtr.__init__(self, statelex, inlex, outlex, name)

    def read(self, fileName, noSelfLoops=False):
        self._g.read_dynamic(fileName, noSelfLoops)


class DecoderFlyWeightPtr(object):
    """decode() pulls every frame of the distribution set's feature stream, scores all distributions on the GPU and runs
    the token-passing kernel; bestHypo() maps the output ids through the output lexicon (decoder.h:748-773)."""

    def __init__(self, dist, beam=100.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silSymbol="SIL-m", eosSymbol="</s>",
                 heapSize=5000, topN=0, generateLattice=True):
        if topN:
            raise K.DsrError(13, "topN > 0 (sorted expansion with per-hypothesis printing) is not supported")
        self._dist = dist; self._cfg = dict(beam=beam, lmScale=lmScale, lmPenalty=lmPenalty, silPenalty=silPenalty)
        self._sil, self._eos = silSymbol, eosSymbol; self._dec = None; self._wfst = None; self._last = None

    def set(self, wfst):
        # _set(): both symbols must exist (decoder.h:740-745 -> jkey_error from List::index)
        silenceX = wfst.inputLexicon().index(self._sil); wfst.outputLexicon().index(self._eos)
        self._dec = K.Decoder(silenceX=silenceX, **self._cfg); self._dec.set(wfst._g); self._wfst = wfst

    def setBeam(self, beam):
        self._cfg["beam"] = beam
        if self._dec:
            self._dec.setBeam(beam)

    def decode(self, verbose=False):
        import torch
        feat = self._dist._cbs.feature()
        rows = [np.array(v, dtype=np.float32) for v in feat]           # __iter__ = reset() + next() until the end
        if not rows:
            raise StopIteration                                         # the exception escapes decode() (decoder.h:691)
        x = torch.from_numpy(np.stack(rows)).cuda()
        sc = self._dist.score_all_frames(x)
        self._last = self._dec.decode_batch(sc[None].contiguous())[0]
        if self._last["status"] != 0:
            raise K.DsrError(self._last["status"], "decode failed")
        return self._last["score"]

    def bestHypo(self, useInputSymbols=False):
        lex = self._wfst.outputLexicon()
        return "".join(lex.symbol(int(w)) + " " for w in self._last["words"])

    def traceBackSucceeded(self):
        return bool(self._last and self._last["reachedFinal"])

    def bestArcs(self):
        return self._last["arcs"]


class DecoderPtr(DecoderFlyWeightPtr):
    """Decoder (decoder.h:1107-1125): the same _Decoder<> search over the dynamic WFSTransducer container."""
