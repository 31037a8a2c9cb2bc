"""btk.dereverberation: SingleChannelWPEDereverberationFeaturePtr, MultiChannelWPEDereverberationPtr, MultiChannelWPEDereverberationFeaturePtr
(dereverberation.i)."""
import ctypes as C

from .. import _capi as K
from .stream import FeatureStreamPtr, lib, _new


class SingleChannelWPEDereverberationFeaturePtr(FeatureStreamPtr):
    def __init__(self, samples, lowerN, upperN, iterationsN=2, loadDb=-20.0, bandWidth=0.0, sampleRate=16000.0,
                 nm="SingleChannelWPEDereverberationFeature"):
        h, _ = _new(lib().dsr_wpe_single_stream_create, samples._h, int(lowerN), int(upperN), int(iterationsN), float(loadDb), float(bandWidth),
                    float(sampleRate), nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(samples,))

    def nextSpeaker(self):
        self.reset()                 # the prediction filters start from zero for every utterance here


class MultiChannelWPEDereverberationPtr:
    """dereverberation.h:89-157: the shared source of the per-channel features.  setInput() adds the channels in order; the prediction
    filters are estimated when the first feature is pulled.  The channel whose feature asks first decides the filter every channel
    of the utterance goes through (dereverberation.cc:381 indexes _Gn with the asking channel inside the loop over channels)."""

    def __init__(self, subbandsN, channelsN, lowerN, upperN, iterationsN=2, loadDb=-20.0, bandWidth=0.0, sampleRate=16000.0):
        self._size, self._channelsN = int(subbandsN), int(channelsN)
        self._cfg = (int(lowerN), int(upperN), int(iterationsN), float(loadDb), float(bandWidth), float(sampleRate))
        if bandWidth > sampleRate / 2.0:
            raise K.DsrError(4, "Bandwidth is greater than the Nyquist rate.")
        self._sources, self._features, self._first = [], [], None

    def size(self):
        return self._size

    def setInput(self, samples):
        if len(self._sources) == self._channelsN:
            raise K.DsrError(2, "Channel capacity exceeded.")              # jallocation_error (dereverberation.cc:359-360)
        self._sources.append(samples)

    def _asked(self, channelX):
        if self._first is None:
            self._first = channelX
            for f in self._features:
                K.check(lib().dsr_wpe_multi_feature_set_filter_channel(f._h, channelX))

    def reset(self):
        self._first = None
        for f in self._features:
            FeatureStreamPtr.reset(f)

    def nextSpeaker(self):
        self.reset()                 # the prediction filters start from zero for every utterance here


class MultiChannelWPEDereverberationFeaturePtr(FeatureStreamPtr):
    def __init__(self, source, channelX, nm="MultiChannelWPEDereverberationFeature"):
        if len(source._sources) != source._channelsN:
            raise K.DsrError(1, "MultiChannelWPEDereverberation: %d of %d inputs are set" % (len(source._sources), source._channelsN))
        arr = (C.c_void_p * source._channelsN)(*[s._h for s in source._sources])
        lo, up, it, ld, bw, sr = source._cfg
        h, _ = _new(lib().dsr_wpe_multi_feature_create, arr, source._channelsN, int(channelX), lo, up, it, ld, bw, sr, nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=tuple(source._sources) + (source,))
        self._source, self._channelX = source, int(channelX)
        source._features.append(self)
        if source._first is not None:
            K.check(lib().dsr_wpe_multi_feature_set_filter_channel(h, source._first))

    def next(self, frameX=-5):
        self._source._asked(self._channelX)
        return FeatureStreamPtr.next(self, frameX)

    def reset(self):
        self._source.reset()
