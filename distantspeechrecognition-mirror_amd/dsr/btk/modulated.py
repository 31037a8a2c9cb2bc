"""btk.modulated: OverSampledDFTAnalysisBankPtr / OverSampledDFTSynthesisBankPtr (modulated.i:117-131,160-164)."""
import ctypes as C

import numpy as np

from .stream import FeatureStreamPtr, lib, _new


class OverSampledDFTAnalysisBankPtr(FeatureStreamPtr):
    def __init__(self, samp, prototype, M, m, r, delayCompensationType=0, nm="OverSampledDFTAnalysisBank"):
        p = np.ascontiguousarray(prototype, dtype=np.float64)
        if p.size != M * m:
            from .. import _capi as K
            raise K.DsrError(4, "Prototype sizes do not match (%d vs. %d)." % (p.size, M * m))
        h, _ = _new(lib().dsr_analysis_bank_create, samp._h, p.ctypes.data_as(C.c_void_p), M, m, r, delayCompensationType, nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(samp,)); self._M = M

    def fftLen(self):
        return self._M


class PerfectReconstructionFFTAnalysisBankPtr(FeatureStreamPtr):
    """modulated.h:377-409: 2M complex bins per frame."""

    def __init__(self, samp, prototype, M, m, r=0, nm="PerfectReconstructionFFTAnalysisBank"):
        p = np.ascontiguousarray(prototype, dtype=np.float64)
        if p.size != 2 * M * m:
            from .. import _capi as K
            raise K.DsrError(4, "Prototype sizes do not match (%d vs. %d)." % (p.size, 2 * M * m))       # modulated.cc:322-324
        h, _ = _new(lib().dsr_pr_analysis_bank_create, samp._h, p.ctypes.data_as(C.c_void_p), M, m, r, nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(samp,)); self._M = M

    def fftLen(self):
        return 2 * self._M

    def nBlocks(self):
        return 4

    def subSampRate(self):
        return 2


class PerfectReconstructionFFTSynthesisBankPtr(FeatureStreamPtr):
    """modulated.h:413-440."""

    def __init__(self, samp, prototype, M, m, r=0, nm="PerfectReconstructionFFTSynthesisBank"):
        p = np.ascontiguousarray(prototype, dtype=np.float64)
        if p.size != 2 * M * m:
            from .. import _capi as K
            raise K.DsrError(4, "Prototype sizes do not match (%d vs. %d)." % (p.size, 2 * M * m))
        h, _ = _new(lib().dsr_pr_synthesis_bank_create, samp._h, p.ctypes.data_as(C.c_void_p), M, m, r, nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(samp,))


class NormalFFTAnalysisBankPtr(FeatureStreamPtr):
    """modulated.i: NormalFFTAnalysisBankPtr(samp, fftLen, r=1, windowType=1, nm=...) (modulated.cc:121-257)."""

    def __init__(self, samp, fftLen, r=1, windowType=1, nm="NormalFFTAnalysisBank"):
        h, _ = _new(lib().dsr_normal_fft_bank_create, samp._h, int(fftLen), int(r), int(windowType), nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(samp,)); self._M = fftLen

    def fftLen(self):
        return self._M


class OverSampledDFTSynthesisBankPtr(FeatureStreamPtr):
    def __init__(self, samp, prototype, M, m, r=0, delayCompensationType=0, gainFactor=1, nm="OverSampledDFTSynthesisBank"):
        p = np.ascontiguousarray(prototype, dtype=np.float64)
        h, _ = _new(lib().dsr_synthesis_bank_create, samp._h, p.ctypes.data_as(C.c_void_p), M, m, r, delayCompensationType, gainFactor, nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(samp,))
