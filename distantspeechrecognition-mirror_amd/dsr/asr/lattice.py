"""asr.lattice: LatticePtr (asr/lattice/lattice.i:79-135) over the C-ABI (include/dsr.h dsr_lattice_*): the lattice a decoder hands back, one
read from a file, or one gathered from another rank, with the reference's operations -- rescore, bestHypo, gammaProbs, prune, pruneEdges, purge,
write / read and the 1-best writers.  Symbols cross the C boundary as indices; the lexica of the constructor turn them into the reference's strings."""
from .. import _capi as K

LogZero = 1.0E10                                              # fsm.h:62


def logAdd(ap, bp):
    """logAdd(LogDouble, LogDouble) (asr/fsm/fsm.cc:38-56, exported by lattice.i:44-45)"""
    import math
    if ap > LogZero:
        raise K.DsrError(K.E_CONSISTENCY, "ap (%g) > LogZero (%g)" % (ap, LogZero))
    if bp > LogZero:
        raise K.DsrError(K.E_CONSISTENCY, "bp (%g) > LogZero (%g)" % (bp, LogZero))
    if ap > bp:
        ap, bp = bp, ap
    return ap - math.log(1.0 + math.exp(ap - bp))


class LatticePtr(object):
    def __init__(self, statelex=None, inlex=None, outlex=None, _lat=None):
        self._state, self._in, self._out = statelex, inlex, outlex
        self._lat = _lat                                      # K.Lattice; None until read() (the reference starts with an initial node and no links)

    def _need(self):
        if self._lat is None:
            raise K.DsrError(K.E_CONSISTENCY, "empty lattice: decode or read one first")
        return self._lat

    def stateLexicon(self):
        return self._state

    def inputLexicon(self):
        return self._in

    def outputLexicon(self):
        return self._out

    def _silX(self, silSymbol):
        if self._in is None:
            raise K.DsrError(K.E_KEY, "no input lexicon to look %s up in" % silSymbol)
        return self._in.index(silSymbol)                      # List::index: jkey_error when absent (mlist.h:109-114)

    def read(self, fileName, noSelfLoops=False, readData=False):
        self._lat = K.Lattice.read(fileName, noSelfLoops, readData, self._in, self._out)

    def write(self, fileName="", useSymbols=False, writeData=False):
        self._need().write(fileName, useSymbols, writeData)

    def rescore(self, lmScale=30.0, lmPenalty=0.0, silPenalty=0.0, silSymbol="SIL-m"):
        return self._need().rescore(lmScale, lmPenalty, silPenalty, self._silX(silSymbol))

    def bestHypo(self, useInputSymbols=False):
        lex = self._in if useInputSymbols else self._out
        ids = self._need().bestHypo(useInputSymbols)
        return "".join(lex.symbol(int(i)) + " " for i in ids)  # the reference prepends "symbol " per link: a trailing blank stays (lattice.cc:292,298)

    def gammaProbs(self, acScale=1.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silSymbol="SIL-m"):
        return self._need().gammaProbs(acScale, lmScale, lmPenalty, silPenalty, self._silX(silSymbol))

    def prune(self, threshold=100.0):
        self._need().prune(threshold)

    def pruneEdges(self, edgesN=0):
        self._need().pruneEdges(edgesN)

    def purge(self):
        self._need().purge()

    def writeCTM(self, conv, channel, spk, utt, cfrom, score, fileName="", frameInterval=0.01, endMarker="</s>"):
        self._need().writeCTM(self._out, conv, channel, spk, utt, cfrom, score, fileName, frameInterval, endMarker)

    def writePhoneCTM(self, conv, channel, spk, utt, cfrom, score, fileName="", frameInterval=0.01, endMarker="</s>"):
        self._need().writePhoneCTM(self._in, conv, channel, spk, utt, cfrom, score, fileName, frameInterval, endMarker)

    def writeHypoHTK(self, conv, channel, spk, utt, cfrom, score, fileName="", flag=0, frameInterval=0.01, endMarker="</s>"):
        self._need().writeHypoHTK(self._out, conv, channel, spk, utt, cfrom, score, fileName, flag, frameInterval, endMarker)

    def writeWordConfs(self, fileName, uttId, endMarker="</s>"):
        self._need().writeWordConfs(self._out, fileName, uttId, endMarker)

    def gammaProbsDist(self, dss, acScale=1.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silSymbol="SIL-m"):
        """lattice.i:119-121: the links' acoustic scores recomputed from the distribution set (scored on the device), then gammaProbs"""
        return self._need().gammaProbsDist(dss._ds, acScale, lmScale, lmPenalty, silPenalty, self._silX(silSymbol))

    def createPhoneLattice(self, *a, **kw):
        raise K.DsrError(K.E_PARAMETER, "createPhoneLattice is not built")

    # pass-throughs used by dist.py and the tests
    def pack(self):
        return self._need().pack()

    def state(self):
        return self._need().state()

    @property
    def data(self):
        return self._need().data

    @property
    def finalStatesN(self):
        return self._need().finalStatesN
