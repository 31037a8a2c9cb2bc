"""btk.feature: the MFCC operator chain with the SWIG constructor signatures of btk/feature/feature.i."""
import ctypes as C

import numpy as np

from .. import _capi as K
from .stream import FeatureStreamPtr, lib, _new


def _b(s):
    return s.encode() if isinstance(s, str) else s


class SampleFeaturePtr(FeatureStreamPtr):
    """feature.i:526-528; read(): see there."""

    def __init__(self, fn="", blockLen=320, shiftLen=160, padZeros=False, nm="Sample"):
        h, _ = _new(lib().dsr_sample_feature_create, blockLen, shiftLen, int(padZeros), _b(nm)); FeatureStreamPtr.__init__(self, h)
        self._rate = 16000
        if fn != "":
            self.read(fn)

    def read(self, fn, format=0, samplerate=16000, chX=1, chN=1, cfrom=0, to=-1, outsamplerate=-1, norm=0.0):
        """SampleFeature::read (feature.cc:243-393, feature.i:487-489).  The reference reads through libsndfile's sf_readf_float; here RIFF/WAV
        PCM of 8, 16, 24 or 32 bits (the library's own reader, dsr_sample_feature_read: the C++ face reads the same way).  norm == 0 keeps the file's integer scale (SFC_SET_NORM_FLOAT off), otherwise samples
        are normalised to [-1, 1) and, for norm != 1, multiplied by norm.  The error branches are the reference's: chX == 0 and chX out of range
        are jconsistency errors, an empty range a jio error.  Sample-rate conversion (outsamplerate != the file's rate; SRCONV builds only) is
        refused.  Returns the number of frames read."""
        n = C.c_int(0)
        st = lib().dsr_sample_feature_read(self._h, _b(fn), int(format), int(samplerate), int(chX), int(chN), int(cfrom), int(to), int(outsamplerate),
                                           C.c_float(norm), C.byref(n))
        if st == K.E_IO:
            raise IOError((lib().dsr_last_error() or b"").decode(errors="replace"))
        K.check(st)
        self._rate = lib().dsr_sample_feature_sample_rate(self._h)
        return n.value

    def setSamples(self, samples, sampleRate=16000):
        a = np.ascontiguousarray(samples, dtype=np.float32)
        K.check(lib().dsr_sample_feature_set_samples(self._h, a.ctypes.data_as(C.c_void_p), a.size, int(sampleRate)))

    def getSampleRate(self):
        return self._rate


def _unary(create, dflt):
    class Op(FeatureStreamPtr):
        def __init__(self, src, *a, **kw):
            nm = kw.pop("nm", dflt)
            h, _ = _new(create, src._h, *self._args(src, *a, **kw), _b(nm)); FeatureStreamPtr.__init__(self, h, keep=(src,))
            if hasattr(self, "_post"):
                self._post()
    return Op


class PreemphasisFeaturePtr(_unary(lambda *a: lib().dsr_preemphasis_create(*a), "Preemphasis")):
    def _args(self, src, mu=0.95):
        return (float(mu),)


class HammingFeaturePtr(_unary(lambda *a: lib().dsr_hamming_create(*a), "Hamming")):
    def _args(self, src):
        return ()


HammingFeatureShortPtr = HammingFeaturePtr


class FFTFeaturePtr(_unary(lambda *a: lib().dsr_fft_create(*a), "FFT")):
    def _args(self, src, fftLen=512):
        return (int(fftLen),)


class SpectralPowerFeaturePtr(_unary(lambda *a: lib().dsr_spectral_power_create(*a), "Power")):
    def _args(self, src, powN=0):
        return (int(powN),)


PowerFeaturePtr = SpectralPowerFeaturePtr


class VTLNFeaturePtr(_unary(lambda *a: lib().dsr_vtln_create(*a), "VTLN")):
    def _args(self, src, coeffN=0, ratio=1.0, edge=1.0, version=1):
        return (int(coeffN), float(ratio), float(edge), int(version))


class MelFeaturePtr(_unary(lambda *a: lib().dsr_mel_create(*a), "MelFFT")):
    def _args(self, src, powN=0, rate=16000.0, low=0.0, up=0.0, filterN=30, version=1):
        return (int(powN), float(rate), float(low), float(up), int(filterN), int(version))


class LogFeaturePtr(_unary(lambda *a: lib().dsr_log_create(*a), "LogMel")):
    def _args(self, src, m=1.0, a=1.0, sphinxFlooring=False):
        return (float(m), float(a), int(sphinxFlooring))


class CepstralFeaturePtr(_unary(lambda *a: lib().dsr_cepstral_create(*a), "Cepstral")):
    def _args(self, src, ncep=13, type=1):
        return (int(ncep), int(type))


class _LpcBase(FeatureStreamPtr):
    """lpc.h:86-195,262-331 (feature.i: WarpMVDRFeaturePtr(src, order=60, correlate=0, warp=0.0, nm=...))."""
    _method = 0; _kind = 0; _dflt = "MVDR"

    def __init__(self, src, order=60, correlate=0, warp=0.0, nm=None):
        h, _ = _new(lib().dsr_lpc_feature_create, src._h, int(order), int(correlate), float(warp), self._method, self._kind, _b(nm or self._dflt))
        FeatureStreamPtr.__init__(self, h, keep=(src,))


class WarpMVDRFeaturePtr(_LpcBase):
    _method = 0; _kind = 0; _dflt = "MVDR"


class BurgMVDRFeaturePtr(_LpcBase):
    _method = 1; _kind = 0; _dflt = "MVDR"


class WarpLPCFeaturePtr(_LpcBase):
    _method = 0; _kind = 1; _dflt = "LPC"


class BurgLPCFeaturePtr(_LpcBase):
    _method = 1; _kind = 1; _dflt = "LPC"


class StorageFeaturePtr(_unary(lambda *a: lib().dsr_storage_create(*a), "Storage")):
    def _args(self, src):
        return ()

    def evaluate(self):
        self.reset(); n = 0
        try:
            while True:
                self.next(n); n += 1
        except StopIteration:
            pass
        return n - 1

    def write(self, fileName, plainText=False):
        """StorageFeature::write (feature.cc:3025-3050)"""
        K.check(lib().dsr_storage_write(self._h, fileName.encode(), int(bool(plainText))))

    def read(self, fileName):
        """StorageFeature::read (feature.cc:3052-3066)"""
        K.check(lib().dsr_storage_read(self._h, fileName.encode()))


class MeanSubtractionFeaturePtr(_unary(lambda *a: lib().dsr_mean_subtraction_create(*a), "Mean Subtraction")):
    def _args(self, src, weight=None, devNormFactor=0.0, runon=False):
        self._weight = weight
        return (float(devNormFactor), int(runon))

    def _post(self):
        if getattr(self, "_weight", None) is not None:                    # (the operator keeps a reference of its own; self._weight keeps the Python face alive)
            K.check(lib().dsr_mean_subtraction_set_weight(self._h, self._weight._h))


class AdjacentFeaturePtr(_unary(lambda *a: lib().dsr_adjacent_create(*a), "Adjacent")):
    def _args(self, src, delta=5):
        return (int(delta),)


class LinearTransformFeaturePtr(_unary(lambda *a: lib().dsr_linear_transform_create(*a), "Transform")):
    def _args(self, src, sz=0):
        self._shape = (int(sz), src.size()); return (int(sz),)

    def setMatrix(self, m):
        a = np.ascontiguousarray(m, dtype=np.float32)
        if a.shape != self._shape:
            raise K.DsrError(5, "Matrix (%d x %d) does not match (%d x %d)" % (a.shape + self._shape))
        K.check(lib().dsr_linear_transform_set(self._h, a.ctypes.data_as(C.c_void_p)))

    def identity(self):
        if self._shape[0] != self._shape[1]:
            raise K.DsrError(5, "Cannot set an (%d x %d) matrix to identity." % self._shape)
        self.setMatrix(np.eye(self._shape[0], dtype=np.float32))

    def load(self, fileName, old=False):
        """LinearTransformFeature::load(fileName, old) (feature.cc:2972-2976): GSL raw float block or Janus 'FMAT' file (gslmatrix.cc:27-96),
        with the reference's error branches (dsr_linear_transform_load)"""
        K.check(lib().dsr_linear_transform_load(self._h, fileName.encode(), int(bool(old))))


class FeatureSetPtr(object):
    """feature.h:1486-1501: name -> stream registry the codebooks look their feature up in."""

    def __init__(self, nm="FeatureSet"):
        self._d = {}

    def add(self, feat):
        self._d[feat.name()] = feat

    def feature(self, nm):
        return self._d[nm]
