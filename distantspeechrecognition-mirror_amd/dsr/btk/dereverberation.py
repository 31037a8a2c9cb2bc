"""btk.dereverberation: SingleChannelWPEDereverberationFeaturePtr (dereverberation.i:67-81)."""
from .stream import FeatureStreamPtr, lib, _new


class SingleChannelWPEDereverberationFeaturePtr(FeatureStreamPtr):
    def __init__(self, samples, lowerN, upperN, iterationsN=2, loadDb=-20.0, bandWidth=0.0, sampleRate=16000.0,
                 nm="SingleChannelWPEDereverberationFeature"):
        h, _ = _new(lib().dsr_wpe_single_stream_create, samples._h, int(lowerN), int(upperN), int(iterationsN), float(loadDb), float(bandWidth),
                    float(sampleRate), nm.encode())
        FeatureStreamPtr.__init__(self, h, keep=(samples,))

    def nextSpeaker(self):
        self.reset()                 # the prediction filters start from zero for every utterance here
