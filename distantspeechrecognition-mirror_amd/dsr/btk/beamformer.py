"""btk.beamformer: SubbandDSPtr / SubbandGSCPtr / SubbandGSCRLSPtr / SubbandMMIPtr / SubbandMVDRPtr (beamformer.i:227-323) as streams."""
import numpy as np

from .. import _capi as K
from .stream import FeatureStreamPtr, lib, _new


class _Subband(FeatureStreamPtr):
    _MODE = 0

    def __init__(self, fftLen=512, halfBandShift=False, nm="SubbandBeamformer"):
        if halfBandShift and self._MODE == 1:
            raise K.DsrError(2, "halfBandShift==true is not yet supported")            # beamformer.cc:2324-2327
        self._fftLen, self._hbs, self._nm = fftLen, halfBandShift, nm
        self._chans = []; self._w = None; FeatureStreamPtr.__init__(self, None)

    def setChannel(self, chan):
        self._chans.append(chan)

    def chanN(self):
        return len(self._chans)

    def _weights(self):
        if self._w is None:
            self._w = K.Beamformer(self._fftLen, len(self._chans), self._hbs); self._w.select(self._MODE)
            h, _ = _new(lib().dsr_subband_bf_create, self._w.h, self._nm.encode()); self._h = h
            for c in self._chans:
                K.check(lib().dsr_subband_bf_set_channel(self._h, c._h))
        return self._w

    def calcArrayManifoldVectors(self, sampleRate, delays):
        self._weights().calcArrayManifoldVectors(sampleRate, delays)

    def next(self, frameX=-5):
        if self._w is None:
            raise K.DsrError(1, "call calcArrayManifoldVectorsX() once")
        return FeatureStreamPtr.next(self, frameX)

    __next__ = next


class SubbandDSPtr(_Subband):
    _MODE = 0


class SubbandGSCPtr(_Subband):
    _MODE = 2

    def calcGSCWeights(self, sampleRate, delaysT):
        self._weights().calcGSCWeights(sampleRate, delaysT)

    def setActiveWeights_f(self, fbinX, packedWeight):
        self._weights().setActiveWeights_f(fbinX, packedWeight)

    def zeroActiveWeights(self):
        self._weights().zeroActiveWeights()


class SubbandGSCRLSPtr(SubbandGSCPtr):
    """beamformer.i:227-253 (SubbandGSCRLS, beamformer.cc:1497-1698).  As in the reference the object keeps adapting across reset(): the
    precision matrices and active weights an utterance leaves are where the next one starts (dsr_bf_rls_carry); only
    initPrecisionMatrix()/setPrecisionMatrix() re-seed them."""

    def __init__(self, fftLen=512, halfBandShift=False, myu=0.9, sigma2=0.01, nm="SubbandGSCRLS"):
        SubbandGSCPtr.__init__(self, fftLen, halfBandShift, nm); self._myu, self._sigma2 = myu, sigma2

    def calcGSCWeights(self, sampleRate, delaysT):
        SubbandGSCPtr.calcGSCWeights(self, sampleRate, delaysT)
        self._weights().rlsConfig(self._myu, self._sigma2)
        self._weights().rlsCarry(True)

    def initPrecisionMatrix(self, sigma2=0.01):
        self._weights().initPrecisionMatrix(sigma2)

    def setPrecisionMatrix(self, fbinX, Pz):
        self._weights().setPrecisionMatrix(fbinX, Pz)

    def setQuadraticConstraint(self, alpha, qctype=1):
        self._weights().setQuadraticConstraint(alpha, qctype)

    def updateActiveWeightVecotrs(self, flag):
        self._weights().updateActiveWeightVecotrs(flag)


class SubbandMMIPtr(_Subband):
    """beamformer.i:255-287 (SubbandMMI, beamformer.cc:1753-2319): one GSC per source, Zelinski post-filter, binary mask."""

    def __init__(self, fftLen=512, halfBandShift=False, targetSourceX=0, nSource=2, pfType=0, alpha=0.9, nm="SubbandMMI"):
        _Subband.__init__(self, fftLen, halfBandShift, nm)
        self._args = (targetSourceX, nSource, pfType, alpha); self._mask = None

    def _weights(self):
        if self._w is None:
            t, n, pf, a = self._args
            self._w = K.SubbandMMI(self._fftLen, len(self._chans), self._hbs, t, n, pf, a)
            if self._mask is not None:
                self._w.useBinaryMask(*self._mask)
            h, _ = _new(lib().dsr_subband_mmi_stream_create, self._w.h, self._fftLen, self._nm.encode()); self._h = h
            for c in self._chans:
                K.check(lib().dsr_subband_bf_set_channel(self._h, c._h))
        return self._w

    def useBinaryMask(self, avgFactor=-1.0, fwidth=1, type=0):
        self._mask = (avgFactor, fwidth, type)
        if self._w is not None:
            self._w.useBinaryMask(avgFactor, fwidth, type)

    def calcWeights(self, sampleRate, delays):
        self._weights().calcWeights(sampleRate, delays)

    def calcWeightsN(self, sampleRate, delays, NC=2):
        self._weights().calcWeightsN(sampleRate, delays, NC)

    def setActiveWeights_f(self, fbinX, packedWeights, option=0):
        if self._w is None:
            raise K.DsrError(1, "call calcWeightsX() once")
        self._w.setActiveWeights_f(fbinX, packedWeights, option)

    def setHiActiveWeights_f(self, fbinX, pkdWa, pkdwb, option=0):
        if self._w is None:
            raise K.DsrError(1, "call calcWeightsX() once")
        self._w.setHiActiveWeights_f(fbinX, pkdWa, pkdwb, option)

    def next(self, frameX=-5):
        if self._w is None:
            raise K.DsrError(1, "call calcWeightsX() once")
        return FeatureStreamPtr.next(self, frameX)

    __next__ = next


class SubbandBlockingMatrixPtr(SubbandGSCPtr):
    """beamformer.h:453-460: a SubbandGSC under another name (its next() is SubbandGSC::next, beamformer.cc:2852-2917)."""


class SubbandMVDRPtr(_Subband):
    _MODE = 1

    def setDiffuseNoiseModel(self, micPositions, sampleRate, sspeed=343740.0):
        self._weights().setDiffuseNoiseModel(micPositions, sampleRate, sspeed); return True

    def divideAllNonDiagonalElements(self, myu):
        self._weights().divideAllNonDiagonalElements(myu)

    def setAllLevelsOfDiagonalLoading(self, w):
        self._weights().setAllLevelsOfDiagonalLoading(w)

    def setNoiseSpatialSpectralMatrix(self, fbinX, Rnn):
        self._weights().setNoiseSpatialSpectralMatrix(fbinX, Rnn); return True

    def calcMVDRWeights(self, sampleRate, dThreshold=1.0e-8, calcInverseMatrix=True):
        self._weights().calcMVDRWeights(sampleRate, dThreshold); return True

    def getMVDRWeights(self, fbinX):
        return self._weights().get(1)[fbinX]


def calcDelaysPolar2(azimuth, elevation, micPositions):
    return K.calcDelaysPolar2(np.float32(azimuth), np.float32(elevation), micPositions)


class SubbandMVDRGSCPtr(SubbandMVDRPtr):
    """beamformer.i (SubbandMVDRGSC, beamformer.h:394-425): setChannel, calcArrayManifoldVectors, noise model, calcMVDRWeights,
    calcBlockingMatrix1/2, setActiveWeights_f; next() = (w_mvdr - B wa)^H X."""
    _MODE = 4

    def setActiveWeights_f(self, fbinX, packedWeight):
        self._weights().setActiveWeights_f(fbinX, packedWeight)

    def zeroActiveWeights(self):
        self._weights().zeroActiveWeights()

    def calcBlockingMatrix1(self, sampleRate, delaysT):
        return self._weights().calcBlockingMatrix1(sampleRate, delaysT)

    def calcBlockingMatrix2(self):
        return self._weights().calcBlockingMatrix2()

    def upgradeBlockingMatrix(self):
        self._weights().upgradeBlockingMatrix()


class SubbandOrthogonalizerPtr(FeatureStreamPtr):
    """beamformer.i (SubbandOrthogonalizer, beamformer.cc:2817-2849): outChanX <= 0 hands the beamformer's output on, outChanX > 0 the output
    of column outChanX-1 of its blocking matrices."""

    def __init__(self, beamformer, outChanX=0, nm="SubbandOrthogonalizer"):
        beamformer._weights()
        h, _ = _new(lib().dsr_subband_orthogonalizer_create, beamformer._h, int(outChanX), nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(beamformer,))
