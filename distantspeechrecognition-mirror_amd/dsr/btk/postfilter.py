"""btk.postfilter: ZelinskiPostFilterPtr (postfilter.i:77-90, postfilter.h:95-126)."""
import ctypes as C

import numpy as np

from .. import _capi as K
from .stream import FeatureStreamPtr, lib, _new

TYPE_ZELINSKI1_REAL, TYPE_ZELINSKI1_ABS, TYPE_APAB, TYPE_ZELINSKI2, NO_USE_POST_FILTER = 0x01, 0x02, 0x04, 0x08, 0x00


class ZelinskiPostFilterPtr(FeatureStreamPtr):
    def __init__(self, output, M, alpha=0.6, type=2, minFrames=0, nm="ZelinskPostFilter"):
        h, _ = _new(lib().dsr_zelinski_stream_create, output._h, int(M), float(alpha), int(type), int(minFrames), nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(output,)); self._M = M; self._type = type; self._chans = []

    def setBeamformer(self, bf):
        """the beamformer's snapshot array (its channel streams) and its weight object (postfilter.cc:376-385,441-444):
        arrayManifold(), or wq() for TYPE_ZELINSKI2 -- both are the delay-and-sum vectors of calcArrayManifoldVectors here"""
        w = bf._weights().get(0)
        for c in bf._chans:
            K.check(lib().dsr_zelinski_stream_set_channel(self._h, c._h)); self._chans.append(c)
        for f in range(self._M // 2 + 1):
            self.setArrayManifoldVector(f, w[f], False)

    def setSnapShotArray(self, channels):
        for c in channels:
            K.check(lib().dsr_zelinski_stream_set_channel(self._h, c._h)); self._chans.append(c)

    def setArrayManifoldVector(self, fbinX, arrayManifoldVector, halfBandShift=False, NC=1):
        if halfBandShift:
            raise K.DsrError(2, "halfBandShift==true is not supported")
        v = np.ascontiguousarray(arrayManifoldVector, np.complex128)
        K.check(lib().dsr_zelinski_stream_set_manifold(self._h, int(fbinX), v.ctypes.data_as(C.c_void_p), v.size))


class McCowanPostFilterPtr(ZelinskiPostFilterPtr):
    """postfilter.i:113-126 (McCowanPostFilter, postfilter.cc:502-945)."""

    def __init__(self, output, fftLen, alpha=0.6, type=2, minFrames=0, threshold=0.99, nm="McCowanPostFilterPtr"):
        h, _ = _new(lib().dsr_mccowan_stream_create, output._h, int(fftLen), float(alpha), int(type), int(minFrames), float(threshold), nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(output,)); self._M = fftLen; self._type = type; self._chans = []; self._C = None

    def _noise(self, what, fbinX, data, chanN, a=0.0, b=0.0):
        d = None if data is None else np.ascontiguousarray(data)
        K.check(lib().dsr_mccowan_stream_set_noise(self._h, what, int(fbinX), None if d is None else d.ctypes.data_as(C.c_void_p), int(chanN), float(a), float(b)))

    def setDiffuseNoiseModel(self, micPositions, sampleRate, sspeed=343740.0):
        mp = np.ascontiguousarray(micPositions, np.float64); self._C = mp.shape[0]
        self._noise(1, 0, mp, self._C, sampleRate, sspeed); return True

    def setNoiseSpatialSpectralMatrix(self, fbinX, Rnn):
        r = np.ascontiguousarray(Rnn, np.complex128); self._C = r.shape[0]
        self._noise(0, fbinX, r, self._C); return True

    def setAllLevelsOfDiagonalLoading(self, diagonalWeight):
        self._noise(2, -1, None, self._C or 0, diagonalWeight)

    def setLevelOfDiagonalLoading(self, fbinX, diagonalWeight):
        self._noise(2, fbinX, None, self._C or 0, diagonalWeight)

    def divideAllNonDiagonalElements(self, myu):
        self._noise(3, 0, None, self._C or 0, myu)


class LefkimmiatisPostFilterPtr(McCowanPostFilterPtr):
    """postfilter.i (LefkimmiatisPostFilter, postfilter.h:180-204, postfilter.cc:948-1210); calcInverseNoiseSpatialSpectralMatrix() is
    implied: the inverse is refreshed whenever the coherence matrices or the manifold change."""

    def __init__(self, output, fftLen, minSV=1.0E-8, fbinX1=0, alpha=0.6, type=2, minFrames=0, threshold=0.99, nm="LefkimmiatisPostFilte"):
        h, _ = _new(lib().dsr_lefkimmiatis_stream_create, output._h, int(fftLen), float(minSV), int(fbinX1), float(alpha), int(type), int(minFrames),
                    float(threshold), nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(output,)); self._M = fftLen; self._type = type; self._chans = []; self._C = None

    def calcInverseNoiseSpatialSpectralMatrix(self):
        return None


class highPassFilterPtr(FeatureStreamPtr):
    """postfilter.i:229-252 (highPassFilter, postfilter.cc:1222-1261)."""

    def __init__(self, output, cutOffFreq, sampleRate, nm="highPassFilter"):
        h, _ = _new(lib().dsr_highpass_filter_create, output._h, float(cutOffFreq), int(sampleRate), nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(output,))
