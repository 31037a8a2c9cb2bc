"""asr.decoder: WFSTFlyWeightPtr / DecoderFlyWeightPtr / DecoderPtr (decoder.i:52-70,147-199) over the C-ABI (include/dsr.h sections 5 and 8)."""
import ctypes as C

import numpy as np

from .. import _capi as K


class WFSTFlyWeightPtr(object):
    def __init__(self, statelex=None, inlex=None, outlex=None, name="WFSTFlyWeight"):
        self._g = K.Wfst(); self._state, self._in, self._out = statelex, inlex, outlex
        h = lambda l: l._h if l is not None else None
        K.check(K.load().dsr_wfst_set_lexicons(self._g.h, h(statelex), h(inlex), h(outlex)))

    def read(self, fileName, binary=False):
        self._g.read(fileName, binary)

    def write(self, fileName, binary=True, useSymbols=False):
        """wfstFlyWeight.cc:415-463; useSymbols: the arcs through the lexica (:499-516)"""
        K.check(K.load().dsr_wfst_write_symbols(self._g.h, fileName.encode(), int(binary), int(useSymbols)))

    def reverse(self, wfst):
        """wfstFlyWeight.cc:141-213"""
        K.check(K.load().dsr_wfst_reverse(self._g.h, wfst._g.h))

    def reverseRead(self, fileName):
        """wfstFlyWeight.cc:215-297"""
        K.check(K.load().dsr_wfst_reverse_read(self._g.h, fileName.encode()))

    def hasFinalState(self):
        return bool(K.load().dsr_wfst_has_final_state(self._g.h))

    def stateLexicon(self):
        return self._state

    def inputLexicon(self):
        return self._in

    def outputLexicon(self):
        return self._out


class WFSTFlyWeightSortedOutputPtr(WFSTFlyWeightPtr):
    """WFSTFlyWeightSortedOutput (asr/decoder/wfstFlyWeight.h:403-424): every node keeps its arcs ordered by (output, input) -- the container
    DecoderWordTrace searches"""

    def __init__(self, statelex=None, inlex=None, outlex=None, name="WFSTFlyWeight"):
        WFSTFlyWeightPtr.__init__(self, statelex, inlex, outlex, name)
        K.check(K.load().dsr_wfst_set_sorted_output(self._g.h, 1))


class WFSTransducerPtr(WFSTFlyWeightPtr):
    """The dynamic container (asr/fsm/fsm.h WFSTransducer): read(fileName, noSelfLoops) per asr/fsm/fsm.cc:901-986.  Node and arc
    order (initial node of its own, arcs prepended, epsilon:epsilon self loops dropped) are those of the fly-weight reader."""

    def __init__(self, statelex=None, inlex=None, outlex=None, name="WFSTransducer"):
        WFSTFlyWeightPtr.__init__(self, statelex, inlex, outlex, name)

    def read(self, fileName, noSelfLoops=False):
        self._g.read_dynamic(fileName, noSelfLoops)


class DistribPath(list):
    """asr/path/distribPath.h:34-60: the names of the distributions along the best path"""


class DecoderFlyWeightPtr(object):
    """DecoderFlyWeight (decoder.h:1127-1139; ctor argument order of decoder.i:191-199).  decode() = dsr_decoder_decode_stream: the feature stream
    of the distribution set is scored and decoded on the device without a host round trip.  heapSize is the bucket count of the reference's token
    hash (decoder.h:48-76) and has no counterpart here (accepted, unused); generateLattice=True keeps the placement log lattice() needs."""

    def __init__(self, dist, beam=100.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silSymbol="SIL-m", eosSymbol="</s>",
                 heapSize=5000, topN=0, generateLattice=True, latticeTokens=1 << 22, maxActive=0):
        self._dist = dist; self._cfg = dict(beam=beam, lmScale=lmScale, lmPenalty=lmPenalty, silPenalty=silPenalty, topN=int(topN), maxActive=maxActive,
                                            latticeTokens=int(latticeTokens) if generateLattice else 0, streams=1)
        self._sil, self._eos = silSymbol, eosSymbol; self._dec = None; self._wfst = None; self._last = None; self._maxPath = 0

    def set(self, wfst):
        # _set(): both symbols must exist (decoder.h:740-745 -> jkey_error from List::index)
        self._dec = K.Decoder(**self._cfg)
        K.check(K.load().dsr_decoder_set_symbols(self._dec.h, wfst._g.h, self._sil.encode(), self._eos.encode()))
        self._dec._g = wfst._g; self._wfst = wfst
        if getattr(self, "_tokLimit", None) is not None:
            K.check(K.load().dsr_decoder_set_token_memory_limit(self._dec.h, self._tokLimit))

    def setBeam(self, beam):
        self._cfg["beam"] = beam
        if self._dec:
            self._dec.setBeam(beam)

    def decode(self, verbose=False):
        if self._dec is None:
            raise K.DsrError(7, "call set() with a transducer first")
        res = K.DecodeResult(); maxPath = 1 << 16
        arcs = np.zeros(maxPath, np.int32); words = np.zeros(maxPath, np.uint32)
        st = K.load().dsr_decoder_decode_stream(self._dec.h, self._dist._ds, C.byref(res), K._ptr(arcs), K._ptr(words), maxPath)
        if st == K.E_ITERATOR:
            raise StopIteration                                         # an empty stream: the exception escapes decode() (decoder.h:691)
        K.check(st)
        self._last = dict(score=res.score, ac=res.ac, lm=res.lm, frames=res.frames, reachedFinal=bool(res.reachedFinal), arcs=arcs[:res.nArcs].copy(),
                          words=words[:res.nWords].copy(), finalStatesN=res.finalStatesN, activeHypos=res.activeHypos)
        if verbose:
            print("Decoded %d frames: score %g, %g active hypotheses per frame" % (res.frames + 1, res.score, res.activeHypos / max(1, res.frames + 1)))
        return res.score

    def _string(self, fn, *a):
        need = C.c_size_t()
        K.check(fn(self._dec.h, 0, *a, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        K.check(fn(self._dec.h, 0, *a, buf, need.value, C.byref(need)))
        return buf.value.decode()

    def bestHypo(self, useInputSymbols=False):
        return self._string(K.load().dsr_decoder_best_hypo, int(bool(useInputSymbols)))

    def bestPath(self):
        need = C.c_size_t(); cnt = C.c_int(); L = K.load()
        K.check(L.dsr_decoder_best_path(self._dec.h, 0, None, 0, C.byref(need), C.byref(cnt)))
        buf = C.create_string_buffer(need.value)
        K.check(L.dsr_decoder_best_path(self._dec.h, 0, buf, need.value, C.byref(need), C.byref(cnt)))
        return DistribPath(buf.value.decode().split("\n")[:cnt.value])

    def finalStatesN(self):
        n = C.c_int(); K.check(K.load().dsr_decoder_final_states_n(self._dec.h, 0, C.byref(n))); return n.value

    def traceBackSucceeded(self):
        ok = C.c_int(); K.check(K.load().dsr_decoder_trace_back_succeeded(self._dec.h, 0, C.byref(ok))); return bool(ok.value)

    def lattice(self):
        """_Decoder::lattice() (decoder.h:805-860): a LatticePtr over the transducer's input lexicon and the decoder's output lexicon"""
        from .lattice import LatticePtr
        lat = self._dec.lattice(0, K.load().dsr_decoder_eos_index(self._dec.h))
        return LatticePtr(None, self._wfst.inputLexicon(), self._wfst.outputLexicon(), _lat=lat)

    def writeGMM(self, conv, channel, spk, utt, cfrom, score, fileName="", frameInterval=0.01):
        """decoder.i:177-178"""
        self._dec.writeGMM(0, conv, channel, spk, utt, cfrom, score, fileName, frameInterval)

    def writeCTM(self, *args, **kw):
        """decoder.h:398-401: the base template CONSTRUCTS a j_error without throwing it -- the shipped call does nothing; so does this one."""
        return None

    def setTokenMemoryLimit(self, limit):
        """decoder.h:396 (no token pool here: the value is kept, see dsr_decoder_set_token_memory_limit)"""
        self._tokLimit = int(limit)
        if self._dec:
            K.check(K.load().dsr_decoder_set_token_memory_limit(self._dec.h, self._tokLimit))

    def bestArcs(self):
        return self._last["arcs"]


class DecoderPtr(DecoderFlyWeightPtr):
    """Decoder (decoder.h:1107-1125): the same _Decoder<> search over the dynamic WFSTransducer container."""


class DecoderWordTracePtr(DecoderFlyWeightPtr):
    """DecoderWordTrace (asr/decoder/decoder.h:1146-1304; decoder.i): the constructor of the reference, set() takes a WFSTFlyWeightSortedOutputPtr.
    generateLattice defaults to True as in the reference -- that search reads the word trace of tokens that have none (decoder.cc:239) and is not
    built: decode() then fails with JCONSISTENCY; with generateLattice=False the 1-best search runs on the device.  bestHypo() returns what the
    shipped class returns (its tokens have no prev(): the last edge's symbol); wordTrace() gives the words along the best token's word traces.
    epsilon / validEndN (early stop, decoder.cc:172-182): epsilon must be 0."""

    def __init__(self, dist, beam=100.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silSymbol="SIL-m", eosSymbol="</s>", heapSize=5000, topN=0,
                 epsilon=0.0, validEndN=30, generateLattice=True, propagateN=5, fastHash=False, insertSilence=False, maxActive=0):
        if epsilon != 0.0:
            raise K.DsrError(13, "DecoderWordTrace: epsilon > 0 (early stop) is not built")
        DecoderFlyWeightPtr.__init__(self, dist, beam, lmScale, lmPenalty, silPenalty, silSymbol, eosSymbol, heapSize, 0, False, 0, maxActive)
        self._cfg.update(wordTrace=1, generateLattice=bool(generateLattice), propagateN=int(propagateN), fastHash=bool(fastHash), insertSilence=bool(insertSilence))

    def wordTrace(self):
        """output-lexicon indices of the words along the best token's word traces, first word first"""
        return [int(w) for w in self._last["words"]]

    def lattice(self):
        raise K.DsrError(4, "Must enable lattice generation during decoding.")                  # decoder.h:807-808
