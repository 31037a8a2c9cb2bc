// host/dsr_streams.hpp -- header-only C++ facade over the C-ABI (include/dsr.h) with the reference's class names, ctor
// argument order, pull protocol and exception types, so SWIG interface files written against btk/stream/stream.h,
// btk/feature/feature.h, btk/modulated/modulated.h and btk/beamformer/beamformer.h keep compiling.
#pragma once
#include <complex>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/dsr.h"

typedef std::string String;

// ---- exceptions (btk/common/jexception.h:41-70)
typedef enum { JERROR, JALLOCATION, JARITHMETIC, JCONSISTENCY, JDIMENSION, JINDEX, JINITIALIZATION, JIO, JITERATOR, JPYTHON,
               JKEY, JNUMERIC, JPARAMETER, JPARSE, JTYPE } error_type;
class j_error : public std::exception {
 public:
  j_error(error_type c, const std::string& w) : _what(w), code(c) {}
  const char* what() const throw() { return _what.c_str(); }
  error_type getCode() { return code; }
 protected:
  std::string _what; error_type code;
};
#define DSR_JERR(cls, val) class cls : public j_error { public: explicit cls(const std::string& w) : j_error(val, w) {} };
DSR_JERR(jallocation_error, JALLOCATION) DSR_JERR(jarithmetic_error, JARITHMETIC) DSR_JERR(jconsistency_error, JCONSISTENCY)
DSR_JERR(jdimension_error, JDIMENSION) DSR_JERR(jindex_error, JINDEX) DSR_JERR(jinitialization_error, JINITIALIZATION)
DSR_JERR(jio_error, JIO) DSR_JERR(jiterator_error, JITERATOR) DSR_JERR(jkey_error, JKEY) DSR_JERR(jnumeric_error, JNUMERIC)
DSR_JERR(jparameter_error, JPARAMETER) DSR_JERR(jparse_error, JPARSE) DSR_JERR(jtype_error, JTYPE)
#undef DSR_JERR

inline void dsr_throw(dsr_status s)
{
  if (s == DSR_OK) return;
  const std::string w = dsr_last_error();
  switch (s - 1) {
    case JALLOCATION: throw jallocation_error(w); case JARITHMETIC: throw jarithmetic_error(w); case JCONSISTENCY: throw jconsistency_error(w);
    case JDIMENSION: throw jdimension_error(w); case JINDEX: throw jindex_error(w); case JINITIALIZATION: throw jinitialization_error(w);
    case JIO: throw jio_error(w); case JITERATOR: throw jiterator_error(w); case JKEY: throw jkey_error(w); case JNUMERIC: throw jnumeric_error(w);
    case JPARAMETER: throw jparameter_error(w); case JPARSE: throw jparse_error(w); case JTYPE: throw jtype_error(w);
    default: throw j_error(JERROR, w);
  }
}

// ---- FeatureStream<item_type> (btk/stream/stream.h:36-75).  The reference's Type is a gsl_vector_X; here next() returns
// a pointer to size() items of item_type living in the operator's own buffer.
template <typename item_type>
class FeatureStream {
 public:
  virtual ~FeatureStream() { if (_h) dsr_stream_release(_h); }
  const String& name() const { return _name; }
  unsigned size() const { return (unsigned) dsr_stream_size(_h); }
  virtual const item_type* next(int frameX = -5) { const void* p = 0; size_t n = 0; dsr_throw(dsr_stream_next(_h, frameX, &p, &n)); return (const item_type*) p; }
  const item_type* current() { const void* p = 0; size_t n = 0; dsr_throw(dsr_stream_current(_h, &p, &n)); return (const item_type*) p; }
  bool isEnd() { return dsr_stream_is_end(_h) != 0; }
  virtual void reset() { dsr_throw(dsr_stream_reset(_h)); }
  virtual int frameX() const { return dsr_stream_frameX(_h); }
  dsr_stream* handle() const { return _h; }
 protected:
  FeatureStream() : _h(0) {}
  void adopt(dsr_stream* h) { _h = h; _name = dsr_stream_name(h); }
  dsr_stream* _h; String _name;
};
typedef FeatureStream<short> VectorShortFeatureStream;
typedef FeatureStream<float> VectorFloatFeatureStream;
typedef FeatureStream<double> VectorFeatureStream;
typedef FeatureStream<std::complex<double> > VectorComplexFeatureStream;
typedef std::shared_ptr<VectorShortFeatureStream> VectorShortFeatureStreamPtr;
typedef std::shared_ptr<VectorFloatFeatureStream> VectorFloatFeatureStreamPtr;
typedef std::shared_ptr<VectorFeatureStream> VectorFeatureStreamPtr;
typedef std::shared_ptr<VectorComplexFeatureStream> VectorComplexFeatureStreamPtr;

#define DSR_OP(cls, base, create_call) { dsr_stream* h = 0; dsr_throw(create_call); this->adopt(h); }

// ---- btk/feature/feature.h
class SampleFeature : public VectorFloatFeatureStream {
 public:
  SampleFeature(const String& fn = "", unsigned blockLen = 320, unsigned shiftLen = 160, bool padZeros = false, const String& nm = "Sample")
  { DSR_OP(SampleFeature, float, dsr_sample_feature_create((int) blockLen, (int) shiftLen, padZeros, nm.c_str(), &h)) if (fn != "") read(fn); }
  // feature.cc:243-393 (what btk/src/superdirectiveBeamformer.cc:150-247 calls): RIFF/WAVE PCM through the library's own reader
  unsigned read(const String& fn, int format = 0, int samplerate = 16000, int chX = 1, int chN = 1, int cfrom = 0, int to = -1, int outsamplerate = -1, float norm = 0.0) {
    int n = 0; dsr_throw(dsr_sample_feature_read(_h, fn.c_str(), format, samplerate, chX, chN, cfrom, to, outsamplerate, norm, &n)); return (unsigned) n;
  }
  int getSampleRate() const { return dsr_sample_feature_sample_rate(_h); }
  void setSamples(const float* samples, size_t n, unsigned sampleRate) { dsr_throw(dsr_sample_feature_set_samples(_h, samples, n, sampleRate)); }
};
class PreemphasisFeature : public VectorFloatFeatureStream {
 public: PreemphasisFeature(const VectorFloatFeatureStreamPtr& samp, double mu, const String& nm = "Preemphasis")
  : _s(samp) { DSR_OP(PreemphasisFeature, float, dsr_preemphasis_create(samp->handle(), mu, nm.c_str(), &h)) }
 private: VectorFloatFeatureStreamPtr _s;
};
class HammingFeature : public VectorFloatFeatureStream {
 public: HammingFeature(const VectorFloatFeatureStreamPtr& samp, const String& nm = "Hamming")
  : _s(samp) { DSR_OP(HammingFeature, float, dsr_hamming_create(samp->handle(), nm.c_str(), &h)) }
 private: VectorFloatFeatureStreamPtr _s;
};
class FFTFeature : public VectorComplexFeatureStream {
 public: FFTFeature(const VectorFloatFeatureStreamPtr& samp, unsigned fftLen, const String& nm = "FFT")
  : _s(samp) { DSR_OP(FFTFeature, cplx, dsr_fft_create(samp->handle(), (int) fftLen, nm.c_str(), &h)) }
 private: VectorFloatFeatureStreamPtr _s;
};
class SpectralPowerFeature : public VectorFeatureStream {
 public: SpectralPowerFeature(const VectorComplexFeatureStreamPtr& fft, unsigned powN = 0, const String& nm = "Power")
  : _s(fft) { DSR_OP(SpectralPowerFeature, double, dsr_spectral_power_create(fft->handle(), (int) powN, nm.c_str(), &h)) }
 private: VectorComplexFeatureStreamPtr _s;
};
class VTLNFeature : public VectorFeatureStream {
 public: VTLNFeature(const VectorFeatureStreamPtr& pow, unsigned coeffN = 0, double ratio = 1.0, double edge = 1.0, int version = 1, const String& nm = "VTLN")
  : _s(pow) { DSR_OP(VTLNFeature, double, dsr_vtln_create(pow->handle(), (int) coeffN, ratio, edge, version, nm.c_str(), &h)) }
 private: VectorFeatureStreamPtr _s;
};
class MelFeature : public VectorFeatureStream {
 public: MelFeature(const VectorFeatureStreamPtr& mag, int powN = 0, float rate = 16000.0, float low = 0.0, float up = 0.0, unsigned filterN = 30, unsigned version = 1, const String& nm = "MelFFT")
  : _s(mag) { DSR_OP(MelFeature, double, dsr_mel_create(mag->handle(), powN, rate, low, up, (int) filterN, (int) version, nm.c_str(), &h)) }
 private: VectorFeatureStreamPtr _s;
};
class LogFeature : public VectorFloatFeatureStream {
 public: LogFeature(const VectorFeatureStreamPtr& mel, double m = 1.0, double a = 1.0, bool sphinxFlooring = false, const String& nm = "LogMel")
  : _s(mel) { DSR_OP(LogFeature, float, dsr_log_create(mel->handle(), m, a, sphinxFlooring, nm.c_str(), &h)) }
 private: VectorFeatureStreamPtr _s;
};
class CepstralFeature : public VectorFloatFeatureStream {
 public: CepstralFeature(const VectorFloatFeatureStreamPtr& mel, unsigned ncep = 13, int type = 1, const String& nm = "Cepstral")
  : _s(mel) { DSR_OP(CepstralFeature, float, dsr_cepstral_create(mel->handle(), (int) ncep, type, nm.c_str(), &h)) }
 private: VectorFloatFeatureStreamPtr _s;
};
class StorageFeature : public VectorFloatFeatureStream {
 public: StorageFeature(const VectorFloatFeatureStreamPtr& src, const String& nm = "Storage")
  : _s(src) { DSR_OP(StorageFeature, float, dsr_storage_create(src->handle(), nm.c_str(), &h)) }
 private: VectorFloatFeatureStreamPtr _s;
};
class MeanSubtractionFeature : public VectorFloatFeatureStream {
 public: MeanSubtractionFeature(const VectorFloatFeatureStreamPtr& src, const VectorFloatFeatureStreamPtr& weight, double devNormFactor = 0.0, bool runon = false, const String& nm = "Mean Subtraction")
  : _s(src), _w(weight) {                                        // the reference's signature (feature.h): weight may be a null pointer
    DSR_OP(MeanSubtractionFeature, float, dsr_mean_subtraction_create(src->handle(), devNormFactor, runon, nm.c_str(), &h))
    if (_w.get()) dsr_throw(dsr_mean_subtraction_set_weight(_h, _w->handle()));
  }
  MeanSubtractionFeature(const VectorFloatFeatureStreamPtr& src, double devNormFactor = 0.0, bool runon = false, const String& nm = "Mean Subtraction")
  : _s(src) { DSR_OP(MeanSubtractionFeature, float, dsr_mean_subtraction_create(src->handle(), devNormFactor, runon, nm.c_str(), &h)) }
 private: VectorFloatFeatureStreamPtr _s, _w;
};
class AdjacentFeature : public VectorFloatFeatureStream {
 public: AdjacentFeature(const VectorFloatFeatureStreamPtr& single, unsigned delta = 5, const String& nm = "Adjacent")
  : _s(single) { DSR_OP(AdjacentFeature, float, dsr_adjacent_create(single->handle(), (int) delta, nm.c_str(), &h)) }
 private: VectorFloatFeatureStreamPtr _s;
};
class LinearTransformFeature : public VectorFloatFeatureStream {
 public: LinearTransformFeature(const VectorFloatFeatureStreamPtr& src, unsigned sz = 0, const String& nm = "Transform")
  : _s(src) { DSR_OP(LinearTransformFeature, float, dsr_linear_transform_create(src->handle(), (int) sz, nm.c_str(), &h)) }
  void setMatrix(const float* m) { dsr_throw(dsr_linear_transform_set(_h, m)); }
 private: VectorFloatFeatureStreamPtr _s;
};

// ---- btk/feature/lpc.h: WarpMVDRFeature, BurgMVDRFeature, WarpLPCFeature, BurgLPCFeature (lpc.h:86-195,262-331)
#define DSR_LPC_OP(NAME, METHOD, KIND, DFLT) class NAME : public VectorFeatureStream { \
 public: NAME(const VectorFloatFeatureStreamPtr& src, unsigned order = 60, unsigned correlate = 0, float warp = 0.0, const String& nm = DFLT) \
  : _s(src) { DSR_OP(NAME, double, dsr_lpc_feature_create(src->handle(), (int) order, (int) correlate, warp, METHOD, KIND, nm.c_str(), &h)) } \
 private: VectorFloatFeatureStreamPtr _s; };
DSR_LPC_OP(WarpMVDRFeature, 0, 0, "MVDR")
DSR_LPC_OP(BurgMVDRFeature, 1, 0, "MVDR")
DSR_LPC_OP(WarpLPCFeature, 0, 1, "LPC")
DSR_LPC_OP(BurgLPCFeature, 1, 1, "LPC")
#undef DSR_LPC_OP

// ---- btk/dereverberation/dereverberation.h
class SingleChannelWPEDereverberationFeature : public VectorComplexFeatureStream {
 public:
  SingleChannelWPEDereverberationFeature(VectorComplexFeatureStreamPtr& samples, unsigned lowerN, unsigned upperN, unsigned iterationsN = 2, double loadDb = -20.0,
                                         double bandWidth = 0.0, double sampleRate = 16000.0, const String& nm = "SingleChannelWPEDereverberationFeature")
  : _s(samples) { DSR_OP(SingleChannelWPEDereverberationFeature, cplx, dsr_wpe_single_stream_create(samples->handle(), (int) lowerN, (int) upperN, (int) iterationsN, loadDb, bandWidth, sampleRate, nm.c_str(), &h)) }
  void nextSpeaker() { reset(); }
 private: VectorComplexFeatureStreamPtr _s;
};

// ---- btk/postfilter/postfilter.h:95-126
class ZelinskiPostFilter : public VectorComplexFeatureStream {
 public:
  ZelinskiPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double alpha = 0.6, int type = 2, int minFrames = 0, const String& nm = "ZelinskPostFilter")
  : _s(output) { DSR_OP(ZelinskiPostFilter, cplx, dsr_zelinski_stream_create(output->handle(), (int) fftLen, alpha, type, minFrames, nm.c_str(), &h)) }
  void setSnapShotChannel(VectorComplexFeatureStreamPtr& chan) { dsr_throw(dsr_zelinski_stream_set_channel(_h, chan->handle())); _chans.push_back(chan); }
  void setArrayManifoldVector(unsigned fbinX, const double* vec, unsigned chanN, bool halfBandShift = false, unsigned NC = 1)
  { (void) halfBandShift; (void) NC; dsr_throw(dsr_zelinski_stream_set_manifold(_h, (int) fbinX, vec, (int) chanN)); }
 private: VectorComplexFeatureStreamPtr _s; std::vector<VectorComplexFeatureStreamPtr> _chans;
};

// ---- btk/postfilter/postfilter.h:128-204.  The noise coherence setters fix the array size at their first call (chanN).
class McCowanPostFilter : public VectorComplexFeatureStream {
 public:
  McCowanPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double alpha = 0.6, int type = 2, int minFrames = 0, float threshold = 0.99f,
                    const String& nm = "McCowanPostFilterPtr")
  : _s(output) { DSR_OP(McCowanPostFilter, cplx, dsr_mccowan_stream_create(output->handle(), (int) fftLen, alpha, type, minFrames, threshold, nm.c_str(), &h)) }
  void setSnapShotChannel(VectorComplexFeatureStreamPtr& chan) { dsr_throw(dsr_zelinski_stream_set_channel(_h, chan->handle())); _chans.push_back(chan); }
  void setArrayManifoldVector(unsigned fbinX, const double* vec, unsigned chanN, bool halfBandShift = false, unsigned NC = 1)
  { (void) halfBandShift; (void) NC; dsr_throw(dsr_zelinski_stream_set_manifold(_h, (int) fbinX, vec, (int) chanN)); }
  void setDiffuseNoiseModel(const double* micPositions /* [chanN][3] */, unsigned chanN, double sampleRate, double sspeed = 343740.0)
  { _C = chanN; dsr_throw(dsr_mccowan_stream_set_noise(_h, 1, 0, micPositions, (int) chanN, sampleRate, sspeed)); }
  void setNoiseSpatialSpectralMatrix(unsigned fbinX, const double* Rnn /* [chanN][chanN] complex */, unsigned chanN)
  { _C = chanN; dsr_throw(dsr_mccowan_stream_set_noise(_h, 0, (int) fbinX, Rnn, (int) chanN, 0.0, 0.0)); }
  void setAllLevelsOfDiagonalLoading(float diagonalWeight) { dsr_throw(dsr_mccowan_stream_set_noise(_h, 2, -1, 0, (int) _C, diagonalWeight, 0.0)); }
  void setLevelOfDiagonalLoading(unsigned fbinX, float diagonalWeight) { dsr_throw(dsr_mccowan_stream_set_noise(_h, 2, (int) fbinX, 0, (int) _C, diagonalWeight, 0.0)); }
  void divideAllNonDiagonalElements(float myu) { dsr_throw(dsr_mccowan_stream_set_noise(_h, 3, 0, 0, (int) _C, myu, 0.0)); }
 protected:
  McCowanPostFilter(VectorComplexFeatureStreamPtr& output) : _s(output), _C(0) {}
  VectorComplexFeatureStreamPtr _s; std::vector<VectorComplexFeatureStreamPtr> _chans; unsigned _C = 0;
};
class LefkimmiatisPostFilter : public McCowanPostFilter {
 public:
  LefkimmiatisPostFilter(VectorComplexFeatureStreamPtr& output, unsigned fftLen, double minSV = 1.0E-8, unsigned fbinX1 = 0, double alpha = 0.6, int type = 2,
                         int minFrames = 0, float threshold = 0.99f, const String& nm = "LefkimmiatisPostFilte")
  : McCowanPostFilter(output) { DSR_OP(LefkimmiatisPostFilter, cplx, dsr_lefkimmiatis_stream_create(output->handle(), (int) fftLen, minSV, (int) fbinX1, alpha, type, minFrames, threshold, nm.c_str(), &h)) }
  void calcInverseNoiseSpatialSpectralMatrix() {}          // implied: refreshed whenever the coherence matrices or the manifold change
};

// ---- btk/dereverberation/dereverberation.h:89-174
class MultiChannelWPEDereverberation {
 public:
  MultiChannelWPEDereverberation(unsigned subbandsN, unsigned channelsN, unsigned lowerN, unsigned upperN, unsigned iterationsN = 2, double loadDb = -20.0,
                                 double bandWidth = 0.0, double sampleRate = 16000.0)
  : _subbandsN(subbandsN), _channelsN(channelsN), _lowerN(lowerN), _upperN(upperN), _iterationsN(iterationsN), _loadDb(loadDb), _bandWidth(bandWidth),
    _sampleRate(sampleRate), _first(-1)
  { if (bandWidth > sampleRate / 2.0) throw jdimension_error("Bandwidth is greater than the Nyquist rate.\n"); }
  unsigned size() const { return _subbandsN; }
  void setInput(VectorComplexFeatureStreamPtr& samples) { if (_sources.size() == _channelsN) throw jallocation_error("Channel capacity exceeded."); _sources.push_back(samples); }
  void reset() { _first = -1; for (size_t i = 0; i < _features.size(); i++) dsr_throw(dsr_stream_reset(_features[i])); }
  void nextSpeaker() { reset(); }
 private:
  friend class MultiChannelWPEDereverberationFeature;
  // the channel whose feature asks for a frame first decides the filter all channels go through (dereverberation.cc:381)
  void asked(int channelX) { if (_first < 0) { _first = channelX; for (size_t i = 0; i < _features.size(); i++) dsr_throw(dsr_wpe_multi_feature_set_filter_channel(_features[i], channelX)); } }
  unsigned _subbandsN, _channelsN, _lowerN, _upperN, _iterationsN; double _loadDb, _bandWidth, _sampleRate; int _first;
  std::vector<VectorComplexFeatureStreamPtr> _sources; std::vector<dsr_stream*> _features;
};
typedef std::shared_ptr<MultiChannelWPEDereverberation> MultiChannelWPEDereverberationPtr;
class MultiChannelWPEDereverberationFeature : public VectorComplexFeatureStream {
 public:
  MultiChannelWPEDereverberationFeature(MultiChannelWPEDereverberationPtr& source, unsigned channelX, const String& nm = "MultiChannelWPEDereverberationFeature")
  : _source(source), _channelX((int) channelX) {
    std::vector<dsr_stream*> in; for (size_t i = 0; i < source->_sources.size(); i++) in.push_back(source->_sources[i]->handle());
    DSR_OP(MultiChannelWPEDereverberationFeature, cplx, dsr_wpe_multi_feature_create(in.data(), (int) in.size(), (int) channelX, (int) source->_lowerN, (int) source->_upperN,
           (int) source->_iterationsN, source->_loadDb, source->_bandWidth, source->_sampleRate, nm.c_str(), &h))
    source->_features.push_back(_h);
    if (source->_first >= 0) dsr_throw(dsr_wpe_multi_feature_set_filter_channel(_h, source->_first));
  }
  const std::complex<double>* next(int frameX = -5) { _source->asked(_channelX); return VectorComplexFeatureStream::next(frameX); }
  void reset() { _source->reset(); }
 private: MultiChannelWPEDereverberationPtr _source; int _channelX;
};

// ---- btk/modulated/modulated.h
class NormalFFTAnalysisBank : public VectorComplexFeatureStream {
 public:
  NormalFFTAnalysisBank(VectorFloatFeatureStreamPtr& samp, unsigned M, unsigned r = 1, unsigned windowType = 1, const String& nm = "NormalFFTAnalysisBank")
  : _s(samp), _M(M) { DSR_OP(NormalFFTAnalysisBank, cplx, dsr_normal_fft_bank_create(samp->handle(), (int) M, (int) r, (int) windowType, nm.c_str(), &h)) }
  unsigned fftLen() const { return _M; }
 private: VectorFloatFeatureStreamPtr _s; unsigned _M;
};
class PerfectReconstructionFFTAnalysisBank : public VectorComplexFeatureStream {
 public:
  PerfectReconstructionFFTAnalysisBank(VectorFloatFeatureStreamPtr& samp, const double* prototype, unsigned M, unsigned m, unsigned r,
                                       const String& nm = "PerfectReconstructionFFTAnalysisBank")
  : _s(samp), _M(M) { DSR_OP(PerfectReconstructionFFTAnalysisBank, cplx, dsr_pr_analysis_bank_create(samp->handle(), prototype, (int) M, (int) m, (int) r, nm.c_str(), &h)) }
  unsigned fftLen() const { return 2 * _M; }
  unsigned nBlocks() const { return 4; }
  unsigned subSampRate() const { return 2; }
 private: VectorFloatFeatureStreamPtr _s; unsigned _M;
};
class PerfectReconstructionFFTSynthesisBank : public VectorFloatFeatureStream {
 public:
  PerfectReconstructionFFTSynthesisBank(VectorComplexFeatureStreamPtr& samp, const double* prototype, unsigned M, unsigned m, unsigned r = 0,
                                        const String& nm = "PerfectReconstructionFFTSynthesisBank")
  : _s(samp) { DSR_OP(PerfectReconstructionFFTSynthesisBank, float, dsr_pr_synthesis_bank_create(samp->handle(), prototype, (int) M, (int) m, (int) r, nm.c_str(), &h)) }
 private: VectorComplexFeatureStreamPtr _s;
};
class OverSampledDFTAnalysisBank : public VectorComplexFeatureStream {
 public:
  OverSampledDFTAnalysisBank(VectorFloatFeatureStreamPtr& samp, const double* prototype, unsigned M, unsigned m, unsigned r,
                             unsigned delayCompensationType = 0, const String& nm = "OverSampledDFTAnalysisBank")
  : _s(samp), _M(M) { DSR_OP(OverSampledDFTAnalysisBank, cplx, dsr_analysis_bank_create(samp->handle(), prototype, (int) M, (int) m, (int) r, (int) delayCompensationType, nm.c_str(), &h)) }
  unsigned fftLen() const { return _M; }
 private: VectorFloatFeatureStreamPtr _s; unsigned _M;
};
class OverSampledDFTSynthesisBank : public VectorFloatFeatureStream {
 public:
  OverSampledDFTSynthesisBank(VectorComplexFeatureStreamPtr& samp, const double* prototype, unsigned M, unsigned m, unsigned r = 0,
                              unsigned delayCompensationType = 0, int gainFactor = 1, const String& nm = "OverSampledDFTSynthesisBank")
  : _s(samp) { DSR_OP(OverSampledDFTSynthesisBank, float, dsr_synthesis_bank_create(samp->handle(), prototype, (int) M, (int) m, (int) r, (int) delayCompensationType, gainFactor, nm.c_str(), &h)) }
 private: VectorComplexFeatureStreamPtr _s;
};

// ---- btk/beamformer/beamformer.h: SubbandDS / SubbandGSC / SubbandMVDR
class SubbandDS : public VectorComplexFeatureStream {
 public:
  SubbandDS(unsigned fftLen = 512, bool halfBandShift = false, const String& nm = "SubbandDS") : _fftLen(fftLen), _hbs(halfBandShift), _nm(nm), _w(0), _mode(0) {}
  ~SubbandDS() { if (_w) dsr_bf_destroy(_w); }
  void setChannel(VectorComplexFeatureStreamPtr& chan) { _channelList.push_back(chan); }
  unsigned chanN() const { return (unsigned) _channelList.size(); }
  void calcArrayManifoldVectors(double sampleRate, const double* delays) { dsr_throw(dsr_bf_calc_array_manifold(weights(), sampleRate, delays)); }
  virtual const std::complex<double>* next(int frameX = -5) {
    if (!_h) throw j_error(JERROR, "call calcArrayManifoldVectorsX() once");
    return VectorComplexFeatureStream::next(frameX);
  }
 protected:
  dsr_bf* weights() {
    if (!_w) {
      dsr_throw(dsr_bf_create((int) _fftLen, (int) chanN(), _hbs, &_w)); dsr_throw(dsr_bf_select(_w, _mode));
      dsr_stream* h = 0; dsr_throw(dsr_subband_bf_create(_w, _nm.c_str(), &h)); adopt(h);
      for (size_t i = 0; i < _channelList.size(); i++) dsr_throw(dsr_subband_bf_set_channel(_h, _channelList[i]->handle()));
    }
    return _w;
  }
  unsigned _fftLen; bool _hbs; String _nm; dsr_bf* _w; int _mode; std::vector<VectorComplexFeatureStreamPtr> _channelList;
};
class SubbandGSC : public SubbandDS {
 public:
  SubbandGSC(unsigned fftLen = 512, bool halfBandShift = false, const String& nm = "SubbandGSC") : SubbandDS(fftLen, halfBandShift, nm) { _mode = 2; }
  void calcGSCWeights(double sampleRate, const double* delaysT) { dsr_throw(dsr_bf_calc_gsc_weights(weights(), sampleRate, delaysT)); }
  void setActiveWeights_f(unsigned fbinX, const double* packedWeight) { dsr_throw(dsr_bf_set_active_weights(weights(), (int) fbinX, packedWeight)); }
  void zeroActiveWeights() { dsr_throw(dsr_bf_zero_active_weights(weights())); }
};
class SubbandMVDR : public SubbandDS {
 public:
  SubbandMVDR(unsigned fftLen = 512, bool halfBandShift = false, const String& nm = "SubbandMVDR") : SubbandDS(fftLen, halfBandShift, nm) {
    if (halfBandShift) throw jallocation_error("halfBandShift==true is not yet supported");
    _mode = 1;
  }
  bool setDiffuseNoiseModel(const double* micPositions, double sampleRate, double sspeed = 343740.0) { dsr_throw(dsr_bf_set_diffuse_noise_model(weights(), micPositions, sampleRate, sspeed)); return true; }
  void divideAllNonDiagonalElements(float myu) { dsr_throw(dsr_bf_divide_nondiagonal(weights(), myu)); }
  void setAllLevelsOfDiagonalLoading(float w) { dsr_throw(dsr_bf_diagonal_loading(weights(), w)); }
  bool calcMVDRWeights(double sampleRate, double dThreshold = 1.0E-8, bool calcInverseMatrix = true) { (void) calcInverseMatrix; dsr_throw(dsr_bf_calc_mvdr_weights(weights(), sampleRate, dThreshold)); return true; }
};
// ---- beamformer.h:264-312 (SubbandMMI): one GSC per source, Zelinski post-filter, binary mask; matrices row-major as gsl_matrix
class SubbandMMI : public VectorComplexFeatureStream {
 public:
  SubbandMMI(unsigned fftLen = 512, bool halfBandShift = false, unsigned targetSourceX = 0, unsigned nSource = 2, int pfType = 0, double alpha = 0.9, const String& nm = "SubbandMMI")
    : _fftLen(fftLen), _hbs(halfBandShift), _target(targetSourceX), _nSource(nSource), _pfType(pfType), _alpha(alpha), _nm(nm), _w(0), _mask(false), _avg(-1.0), _fwidth(1), _mtype(0) {}
  ~SubbandMMI() { if (_w) dsr_mmi_destroy(_w); }
  void setChannel(VectorComplexFeatureStreamPtr& chan) { _channelList.push_back(chan); }
  unsigned chanN() const { return (unsigned) _channelList.size(); }
  void useBinaryMask(double avgFactor = -1.0, unsigned fwidth = 1, unsigned type = 0) { _mask = true; _avg = avgFactor; _fwidth = fwidth; _mtype = type; if (_w) dsr_throw(dsr_mmi_use_binary_mask(_w, avgFactor, fwidth, type)); }
  void calcWeights(double sampleRate, const double* delays /*[nSource][chanN]*/) { dsr_throw(dsr_mmi_calc_weights(weights(), sampleRate, delays)); }
  void calcWeightsN(double sampleRate, const double* delays, unsigned NC = 2) { dsr_throw(dsr_mmi_calc_weights_n(weights(), sampleRate, delays, NC)); }
  void setActiveWeights_f(unsigned fbinX, const double* packedWeights, size_t rows, size_t cols, int option = 0) {
    if (!_w) throw j_error(JERROR, "call calcWeightsX() once");
    dsr_throw(dsr_mmi_set_active_weights_f(_w, fbinX, packedWeights, rows, cols, option));
  }
  void setHiActiveWeights_f(unsigned fbinX, const double* pkdWa, size_t nWa, const double* pkdwb, size_t nWb, int option = 0) {
    if (!_w) throw j_error(JERROR, "call calcWeightsX() once");
    dsr_throw(dsr_mmi_set_hi_active_weights_f(_w, fbinX, pkdWa, nWa, pkdwb, nWb, option));
  }
  virtual const std::complex<double>* next(int frameX = -5) {
    if (!_h) throw j_error(JERROR, "call calcWeightsX() once");
    return VectorComplexFeatureStream::next(frameX);
  }
 private:
  dsr_mmi* weights() {
    if (!_w) {
      dsr_throw(dsr_mmi_create((int) _fftLen, (int) chanN(), _hbs, (int) _target, (int) _nSource, _pfType, _alpha, &_w));
      if (_mask) dsr_throw(dsr_mmi_use_binary_mask(_w, _avg, _fwidth, _mtype));
      dsr_stream* h = 0; dsr_throw(dsr_subband_mmi_stream_create(_w, (int) _fftLen, _nm.c_str(), &h)); adopt(h);
      for (size_t i = 0; i < _channelList.size(); i++) dsr_throw(dsr_subband_bf_set_channel(_h, _channelList[i]->handle()));
    }
    return _w;
  }
  unsigned _fftLen; bool _hbs; unsigned _target, _nSource; int _pfType; double _alpha; String _nm; dsr_mmi* _w;
  bool _mask; double _avg; unsigned _fwidth, _mtype; std::vector<VectorComplexFeatureStreamPtr> _channelList;
};
#undef DSR_OP

// ======================================================================================================================
// ASR side (asr/dictionary, asr/gaussian, asr/decoder): the classes decoder.i:52-199 and gaussian.i:281-283,465-467 wrap, with
// their constructor argument order, over include/dsr.h sections 4, 5 and 8.
// ======================================================================================================================
#include <fstream>
#include <map>
#include <sstream>

// ---- btk/feature/feature.h:1486-1501
class FeatureSet {
 public:
  explicit FeatureSet(const String& nm = "FeatureSet") : _name(nm) {}
  const String& name() const { return _name; }
  void add(VectorFloatFeatureStreamPtr& feat) { _list[feat->name()] = feat; }
  VectorFloatFeatureStreamPtr& feature(const String& nm) {
    std::map<String, VectorFloatFeatureStreamPtr>::iterator it = _list.find(nm);
    if (it == _list.end()) throw jkey_error("Could not find key " + nm + " in list " + _name);          // List::operator[] (mlist.h:109-114)
    return it->second;
  }
 private:
  String _name; std::map<String, VectorFloatFeatureStreamPtr> _list;
};
typedef std::shared_ptr<FeatureSet> FeatureSetPtr;

// ---- asr/dictionary/distribTree.h:40-65
class Lexicon {
 public:
  explicit Lexicon(const String& nm, const String& fileName = "") : _h(0) { dsr_throw(dsr_lexicon_create(nm.c_str(), fileName.c_str(), &_h)); }
  ~Lexicon() { dsr_lexicon_destroy(_h); }
  void clear() { dsr_throw(dsr_lexicon_clear(_h)); }
  String name() const { return dsr_lexicon_name(_h); }
  unsigned size() const { return (unsigned) dsr_lexicon_size(_h); }
  void read(const String& fileName) { dsr_throw(dsr_lexicon_read(_h, fileName.c_str())); }
  void write(const String& fileName, bool writeHeader = false) const { dsr_throw(dsr_lexicon_write(_h, fileName.c_str(), writeHeader)); }
  unsigned index(const String& symbol, bool create = false) { unsigned i = 0; dsr_throw(dsr_lexicon_index(_h, symbol.c_str(), create, &i)); return i; }
  String symbol(unsigned idx) const { const char* p = 0; dsr_throw(dsr_lexicon_symbol(_h, idx, &p)); return p; }
  bool isPresent(const String& symbol) const { return dsr_lexicon_is_present(_h, symbol.c_str()) != 0; }
  dsr_lexicon* handle() const { return _h; }
 private:
  Lexicon(const Lexicon&); Lexicon& operator=(const Lexicon&);
  dsr_lexicon* _h;
};
typedef std::shared_ptr<Lexicon> LexiconPtr;

// ---- asr/gaussian: CodebookSetBasic(descFile, fs, cbkFile) (gaussian.i:281-283; description lines "name featureName refN dimN covType",
// ';' comments, codebookBasic.cc:804-828) and DistribSetBasic(cbs, descFile, distFile) (gaussian.i:465-467; lines "name codebookName",
// distribBasic.cc:218-232).  The big-endian set files are read by dsr_gmm_load.
inline std::vector<std::vector<String> > dsr_desc_rows(const String& path)
{
  std::vector<std::vector<String> > rows; if (path.empty()) return rows;
  std::ifstream f(path.c_str()); if (!f) throw jio_error("Could not open file " + path);
  String line;
  while (std::getline(f, line)) { if (!line.empty() && line[0] == ';') continue; std::istringstream is(line); std::vector<String> t; String w; while (is >> w) t.push_back(w); if (!t.empty()) rows.push_back(t); }
  return rows;
}
class CodebookSetBasic {
 public:
  CodebookSetBasic(const String& descFile = "", FeatureSetPtr fs = FeatureSetPtr(), const String& cbkFile = "") : _fs(fs), _cbkFile(cbkFile), _desc(dsr_desc_rows(descFile)) {}
  unsigned ncbks() const { return (unsigned) _desc.size(); }
  const String& cbkFile() const { return _cbkFile; }
  VectorFloatFeatureStreamPtr& feature() {
    if (_desc.empty() || _desc[0].size() < 2 || !_fs) throw jconsistency_error("the codebook description names no feature");
    return _fs->feature(_desc[0][1]);
  }
 private:
  FeatureSetPtr _fs; String _cbkFile; std::vector<std::vector<String> > _desc;
};
typedef std::shared_ptr<CodebookSetBasic> CodebookSetBasicPtr;

class DistribSetBasic;
class DistribBasic {                                          // Distrib::score(frameX), name() (distribBasic.h:40-60)
 public:
  DistribBasic(dsr_distribset* ds, int x) : _ds(ds), _x(x) {}
  float score(int frameX) { float s = 0.f; dsr_throw(dsr_distribset_score(_ds, _x, frameX, &s)); return s; }
  String name() const { return dsr_distribset_name(_ds, _x); }
 private:
  dsr_distribset* _ds; int _x;
};
class DistribSetBasic {
 public:
  DistribSetBasic(CodebookSetBasicPtr& cbs, const String& descFile = "", const String& distFile = "") : _cbs(cbs), _gmm(0), _ds(0) {
    const std::vector<std::vector<String> > d = dsr_desc_rows(descFile);
    dsr_throw(dsr_gmm_load(cbs->cbkFile().c_str(), distFile.c_str(), &_gmm));
    if (!d.empty() && (int) d.size() != dsr_gmm_num_dists(_gmm)) { dsr_gmm_destroy(_gmm); throw jconsistency_error("the description and the file hold different numbers of distributions"); }
    _feat = cbs->feature();
    const dsr_status s = dsr_distribset_create(_gmm, _feat->handle(), 0, &_ds);
    if (s != DSR_OK) { dsr_gmm_destroy(_gmm); dsr_throw(s); }
  }
  ~DistribSetBasic() { dsr_distribset_destroy(_ds); dsr_gmm_destroy(_gmm); }
  unsigned ndists() const { return (unsigned) dsr_distribset_ndists(_ds); }
  unsigned index(const String& key) const { int x = 0; dsr_throw(dsr_distribset_find(_ds, key.c_str(), &x)); return (unsigned) x; }
  DistribBasic find(const String& key) { return DistribBasic(_ds, (int) index(key)); }
  DistribBasic find(unsigned dsX) { if (dsX >= ndists()) throw jindex_error("distribution index out of range"); return DistribBasic(_ds, (int) dsX); }
  void resetCache() { dsr_throw(dsr_distribset_reset_cache(_ds)); }
  void resetFeature() { dsr_throw(dsr_distribset_reset_feature(_ds)); }
  dsr_distribset* handle() const { return _ds; }
 private:
  DistribSetBasic(const DistribSetBasic&); DistribSetBasic& operator=(const DistribSetBasic&);
  CodebookSetBasicPtr _cbs; VectorFloatFeatureStreamPtr _feat; dsr_gmm* _gmm; dsr_distribset* _ds;
};
typedef std::shared_ptr<DistribSetBasic> DistribSetBasicPtr;
typedef DistribSetBasicPtr DistribSetPtr;

// ---- asr/decoder: WFSTFlyWeight(statelex, inlex, outlex, name) (decoder.i:52-70)
class WFSTFlyWeight {
 public:
  WFSTFlyWeight(LexiconPtr& statelex, LexiconPtr& inlex, LexiconPtr& outlex, const String& name = "WFSTFlyWeight")
    : _stateLexicon(statelex), _inputLexicon(inlex), _outputLexicon(outlex), _name(name), _h(0) {
    dsr_throw(dsr_wfst_create(&_h));
    dsr_throw(dsr_wfst_set_lexicons(_h, statelex ? statelex->handle() : 0, inlex ? inlex->handle() : 0, outlex ? outlex->handle() : 0));
  }
  ~WFSTFlyWeight() { dsr_wfst_destroy(_h); }
  void read(const String& fileName, bool binary = false) { dsr_throw(dsr_wfst_read(_h, fileName.c_str(), binary)); }
  void write(const String& fileName, bool binary = true, bool useSymbols = false) { dsr_throw(dsr_wfst_write_symbols(_h, fileName.c_str(), binary, useSymbols)); }
  void reverse(const std::shared_ptr<WFSTFlyWeight>& wfst) { dsr_throw(dsr_wfst_reverse(_h, wfst->handle())); }      // wfstFlyWeight.cc:141-213
  void reverseRead(const String& fileName) { dsr_throw(dsr_wfst_reverse_read(_h, fileName.c_str())); }                // :215-297
  bool hasFinalState() const { return dsr_wfst_has_final_state(_h) != 0; }
  LexiconPtr& stateLexicon() { return _stateLexicon; }
  LexiconPtr& inputLexicon() { return _inputLexicon; }
  LexiconPtr& outputLexicon() { return _outputLexicon; }
  dsr_wfst* handle() const { return _h; }
 private:
  WFSTFlyWeight(const WFSTFlyWeight&); WFSTFlyWeight& operator=(const WFSTFlyWeight&);
  LexiconPtr _stateLexicon, _inputLexicon, _outputLexicon; String _name; dsr_wfst* _h;
};
typedef std::shared_ptr<WFSTFlyWeight> WFSTFlyWeightPtr;
// asr/decoder/wfstFlyWeight.h:403-424: every node keeps its arcs ordered by (output, input); what DecoderWordTrace::set takes
class WFSTFlyWeightSortedOutput : public WFSTFlyWeight {
 public:
  WFSTFlyWeightSortedOutput(LexiconPtr& statelex, LexiconPtr& inlex, LexiconPtr& outlex, const String& name = "WFSTFlyWeight")
    : WFSTFlyWeight(statelex, inlex, outlex, name) { dsr_throw(dsr_wfst_set_sorted_output(handle(), 1)); }
};
typedef std::shared_ptr<WFSTFlyWeightSortedOutput> WFSTFlyWeightSortedOutputPtr;

typedef std::vector<String> DistribPath;                       // asr/path/distribPath.h:34-60: the distribution names along a path
// asr/lattice Lattice(statelex, inlex, outlex) (lattice.i:79-135, lattice.h:188-330): what _Decoder::lattice() returns, or an object to read() into
class Lattice {
 public:
  Lattice(LexiconPtr statelex, LexiconPtr inlex, LexiconPtr outlex) : _stateLexicon(statelex), _inputLexicon(inlex), _outputLexicon(outlex), _h(0) {}
  Lattice(dsr_lattice* h, LexiconPtr inlex, LexiconPtr outlex) : _inputLexicon(inlex), _outputLexicon(outlex), _h(h) {}
  explicit Lattice(dsr_lattice* h) : _h(h) {}
  ~Lattice() { if (_h) dsr_lattice_destroy(_h); }
  void read(const String& fileName, bool noSelfLoops = false, bool readData = false) {
    dsr_lattice* n = 0;
    dsr_throw(dsr_lattice_read(fileName.c_str(), noSelfLoops, readData, _inputLexicon ? _inputLexicon->handle() : 0, _outputLexicon ? _outputLexicon->handle() : 0, &n));
    if (_h) dsr_lattice_destroy(_h);
    _h = n;
  }
  void write(const String& fileName = "", bool useSymbols = false, bool writeData = false) {
    if (useSymbols) throw jparameter_error("useSymbols: write the numeric form and map the symbols with the lexica");
    dsr_throw(dsr_lattice_write(need(), fileName.c_str(), writeData));
  }
  float rescore(double lmScale = 30.0, double lmPenalty = 0.0, double silPenalty = 0.0, const String& silSymbol = "SIL-m") {
    float s = 0.0f; dsr_throw(dsr_lattice_rescore(need(), lmScale, lmPenalty, silPenalty, silX(silSymbol), &s)); return s;
  }
  String bestHypo(bool useInputSymbols = false) {
    int n = 0; dsr_throw(dsr_lattice_best_hypo(need(), useInputSymbols, 0, 0, &n));
    std::vector<uint32_t> ids((size_t) (n > 0 ? n : 1)); dsr_throw(dsr_lattice_best_hypo(_h, useInputSymbols, ids.data(), (int) ids.size(), &n));
    LexiconPtr& lex = useInputSymbols ? _inputLexicon : _outputLexicon;
    if (!lex) throw jkey_error("no lexicon to name the symbols with");
    String hypo; for (int i = 0; i < n; i++) hypo += lex->symbol(ids[(size_t) i]) + " ";           // "symbol " per link, as lattice.cc:292,298 build it
    return hypo;
  }
  double gammaProbs(double acScale = 1.0, double lmScale = 12.0, double lmPenalty = 0.0, double silPenalty = 0.0, const String& silSymbol = "SIL-m") {
    double p = 0.0; dsr_throw(dsr_lattice_gamma_probs(need(), acScale, lmScale, lmPenalty, silPenalty, silX(silSymbol), &p)); return p;
  }
  double gammaProbsDist(DistribSetBasicPtr& dss, double acScale = 1.0, double lmScale = 12.0, double lmPenalty = 0.0, double silPenalty = 0.0, const String& silSymbol = "SIL-m") {
    double p = 0.0; dsr_throw(dsr_lattice_gamma_probs_dist(need(), dss->handle(), acScale, lmScale, lmPenalty, silPenalty, silX(silSymbol), &p)); return p;
  }
  void prune(double threshold = 100.0) { dsr_throw(dsr_lattice_prune(need(), threshold)); }
  void pruneEdges(unsigned edgesN = 0) { dsr_throw(dsr_lattice_prune_edges(need(), edgesN)); }
  void purge() { dsr_throw(dsr_lattice_purge(need())); }
  void writeCTM(const String& conv, const String& channel, const String& spk, const String& utt, double cfrom, double score, const String& fileName = "",
                double frameInterval = 0.01, const String& endMarker = "</s>") {
    dsr_throw(dsr_lattice_write_ctm(need(), lexh(_outputLexicon), conv.c_str(), channel.c_str(), spk.c_str(), utt.c_str(), cfrom, score, fileName.c_str(), frameInterval, endMarker.c_str()));
  }
  void writePhoneCTM(const String& conv, const String& channel, const String& spk, const String& utt, double cfrom, double score, const String& fileName = "",
                     double frameInterval = 0.01, const String& endMarker = "</s>") {
    dsr_throw(dsr_lattice_write_phone_ctm(need(), lexh(_inputLexicon), conv.c_str(), channel.c_str(), spk.c_str(), utt.c_str(), cfrom, score, fileName.c_str(), frameInterval, endMarker.c_str()));
  }
  void writeHypoHTK(const String& conv, const String& channel, const String& spk, const String& utt, double cfrom, double score, const String& fileName = "",
                    int flag = 0, double frameInterval = 0.01, const String& endMarker = "</s>") {
    dsr_throw(dsr_lattice_write_hypo_htk(need(), lexh(_outputLexicon), conv.c_str(), channel.c_str(), spk.c_str(), utt.c_str(), cfrom, score, fileName.c_str(), flag, frameInterval, endMarker.c_str()));
  }
  void writeWordConfs(const String& fileName, const String& uttId, const String& endMarker = "</s>") {
    dsr_throw(dsr_lattice_write_word_confs(need(), lexh(_outputLexicon), fileName.c_str(), uttId.c_str(), endMarker.c_str()));
  }
  unsigned nodesN() const { return _h ? (unsigned) dsr_lattice_num_nodes(_h) : 0u; }
  unsigned edgesN() const { return _h ? (unsigned) dsr_lattice_num_edges(_h) : 0u; }
  LexiconPtr& stateLexicon() { return _stateLexicon; }
  LexiconPtr& inputLexicon() { return _inputLexicon; }
  LexiconPtr& outputLexicon() { return _outputLexicon; }
  dsr_lattice* handle() const { return _h; }
 private:
  Lattice(const Lattice&); Lattice& operator=(const Lattice&);
  dsr_lattice* need() const { if (!_h) throw jconsistency_error("empty lattice: decode or read one first"); return _h; }
  static dsr_lexicon* lexh(const LexiconPtr& l) { if (!l) throw jkey_error("no lexicon to name the symbols with"); return l->handle(); }
  unsigned silX(const String& silSymbol) { if (!_inputLexicon) throw jkey_error("no input lexicon to look " + silSymbol + " up in"); return _inputLexicon->index(silSymbol); }
  LexiconPtr _stateLexicon, _inputLexicon, _outputLexicon; dsr_lattice* _h;
};
typedef std::shared_ptr<Lattice> LatticePtr;

// DecoderFlyWeight(dist, beam, lmScale, lmPenalty, silPenalty, silSymbol, eosSymbol, heapSize, topN, generateLattice) (decoder.i:147-199).
// heapSize sizes the reference's token hash (decoder.h:48-76) and has no counterpart here.
class DecoderFlyWeight {
 public:
  DecoderFlyWeight(DistribSetPtr& dist, double beam = 100.0, double lmScale = 12.0, double lmPenalty = 0.0, double silPenalty = 0.0,
                   const String& silSymbol = "SIL-m", const String& eosSymbol = "</s>", unsigned heapSize = 5000, unsigned topN = 0, bool generateLattice = true)
    : _dist(dist), _sil(silSymbol), _eos(eosSymbol), _h(0), _score(0.0) {
    (void) heapSize;
    dsr_decoder_cfg c; dsr_decoder_default_cfg(&c);
    c.beam = beam; c.lmScale = lmScale; c.lmPenalty = lmPenalty; c.silPenalty = silPenalty; c.topN = (int) topN; c.streams = 1;
    c.latticeTokens = generateLattice ? ((int64_t) 1 << 22) : 0;
    dsr_throw(dsr_decoder_create(&c, &_h));
  }
  ~DecoderFlyWeight() { dsr_decoder_destroy(_h); }
  void set(WFSTFlyWeightPtr& wfst) { dsr_throw(dsr_decoder_set_symbols(_h, wfst->handle(), _sil.c_str(), _eos.c_str())); _wfst = wfst; }
  double decode(bool verbose = false) {
    (void) verbose;
    dsr_decode_result r; std::vector<int32_t> arcs((size_t) 1 << 16); std::vector<uint32_t> words((size_t) 1 << 16);
    dsr_throw(dsr_decoder_decode_stream(_h, _dist->handle(), &r, arcs.data(), words.data(), (int) arcs.size()));
    _score = r.score; return _score;
  }
  String bestHypo(bool useInputSymbols = false) {
    size_t need = 0; dsr_throw(dsr_decoder_best_hypo(_h, 0, useInputSymbols, 0, 0, &need));
    std::vector<char> b(need); dsr_throw(dsr_decoder_best_hypo(_h, 0, useInputSymbols, b.data(), b.size(), &need)); return String(b.data());
  }
  DistribPath bestPath() {
    size_t need = 0; int n = 0; dsr_throw(dsr_decoder_best_path(_h, 0, 0, 0, &need, &n));
    std::vector<char> b(need); dsr_throw(dsr_decoder_best_path(_h, 0, b.data(), b.size(), &need, &n));
    DistribPath p; std::istringstream is(String(b.data())); String w; while (std::getline(is, w)) p.push_back(w); return p;
  }
  unsigned finalStatesN() const { int n = 0; dsr_throw(dsr_decoder_final_states_n(_h, 0, &n)); return (unsigned) n; }
  bool traceBackSucceeded() const { int ok = 0; dsr_throw(dsr_decoder_trace_back_succeeded(_h, 0, &ok)); return ok != 0; }
  LatticePtr lattice() {
    dsr_lattice* l = 0; dsr_throw(dsr_decoder_lattice(_h, 0, dsr_decoder_eos_index(_h), &l));
    return LatticePtr(_wfst ? new Lattice(l, _wfst->inputLexicon(), _wfst->outputLexicon()) : new Lattice(l));
  }
  void writeGMM(const String& conv, const String& channel, const String& spk, const String& utt, double cfrom, double score, const String& fileName = "", double frameInterval = 0.01) {
    dsr_throw(dsr_decoder_write_gmm(_h, 0, conv.c_str(), channel.c_str(), spk.c_str(), utt.c_str(), cfrom, score, fileName.c_str(), frameInterval));
  }
  // decoder.h:398-401: the base template constructs a j_error and does not throw it -- a silent no-op in the shipped code, and here
  void writeCTM(const String&, const String&, const String&, const String&, double, double, const String& = "", double = 0.01) {}
  void setTokenMemoryLimit(unsigned limit) { dsr_throw(dsr_decoder_set_token_memory_limit(_h, limit)); }           // decoder.h:396
  void setBeam(double beam) { dsr_throw(dsr_decoder_set_beam(_h, beam)); }
  dsr_decoder* handle() const { return _h; }
 private:
  DecoderFlyWeight(const DecoderFlyWeight&); DecoderFlyWeight& operator=(const DecoderFlyWeight&);
  DistribSetPtr _dist; WFSTFlyWeightPtr _wfst; String _sil, _eos; dsr_decoder* _h; double _score;
};
typedef std::shared_ptr<DecoderFlyWeight> DecoderFlyWeightPtr;

// asr/decoder/decoder.h:1146-1304 (constructor :1227-1232).  generateLattice defaults to true as in the reference; that search reads the word trace of tokens
// that have none (decoder.cc:239) and is not built: decode() then throws jconsistency_error.  With generateLattice = false the 1-best search runs on the device.
class DecoderWordTrace {
 public:
  DecoderWordTrace(DistribSetPtr& dist, double beam = 100.0, double lmScale = 12.0, double lmPenalty = 0.0, double silPenalty = 0.0, const String& silSymbol = "SIL-m",
                   const String& eosSymbol = "</s>", unsigned heapSize = 5000, unsigned topN = 0, double epsilon = 0.0, unsigned validEndN = 30,
                   bool generateLattice = true, unsigned propagateN = 5, bool fastHash = false, bool insertSilence = false)
    : _dist(dist), _sil(silSymbol), _eos(eosSymbol), _h(0) {
    (void) heapSize; (void) topN; (void) validEndN;
    if (epsilon != 0.0) throw j_error(JPARAMETER, "DecoderWordTrace: epsilon > 0 (early stop, decoder.cc:172-182) is not built");
    dsr_decoder_cfg c; dsr_decoder_default_cfg(&c);
    c.beam = beam; c.lmScale = lmScale; c.lmPenalty = lmPenalty; c.silPenalty = silPenalty; c.streams = 1;
    c.wordTrace = 1; c.wordTraceLattice = generateLattice; c.propagateN = (int) propagateN; c.fastHash = fastHash; c.insertSilence = insertSilence;
    dsr_throw(dsr_decoder_create(&c, &_h));
  }
  ~DecoderWordTrace() { dsr_decoder_destroy(_h); }
  void set(WFSTFlyWeightSortedOutputPtr& wfst) { dsr_throw(dsr_decoder_set_symbols(_h, wfst->handle(), _sil.c_str(), _eos.c_str())); _wfst = wfst; }
  double decode(bool verbose = false) {
    (void) verbose;
    dsr_decode_result r; std::vector<int32_t> arcs(16); _words.assign((size_t) 1 << 16, 0u);
    dsr_throw(dsr_decoder_decode_stream(_h, _dist->handle(), &r, arcs.data(), _words.data(), (int) _words.size()));
    _words.resize((size_t) r.nWords); return r.score;
  }
  String bestHypo(bool useInputSymbols = false) {                       // (the shipped class's tokens have no prev(): one symbol)
    size_t need = 0; dsr_throw(dsr_decoder_best_hypo(_h, 0, useInputSymbols, 0, 0, &need));
    std::vector<char> b(need); dsr_throw(dsr_decoder_best_hypo(_h, 0, useInputSymbols, b.data(), b.size(), &need)); return String(b.data());
  }
  const std::vector<uint32_t>& wordTrace() const { return _words; }     // the words along the best token's word traces
  unsigned finalStatesN() const { int n = 0; dsr_throw(dsr_decoder_final_states_n(_h, 0, &n)); return (unsigned) n; }
  bool traceBackSucceeded() const { int ok = 0; dsr_throw(dsr_decoder_trace_back_succeeded(_h, 0, &ok)); return ok != 0; }
  void setBeam(double beam) { dsr_throw(dsr_decoder_set_beam(_h, beam)); }
 private:
  DecoderWordTrace(const DecoderWordTrace&); DecoderWordTrace& operator=(const DecoderWordTrace&);
  DistribSetPtr _dist; WFSTFlyWeightSortedOutputPtr _wfst; String _sil, _eos; dsr_decoder* _h; std::vector<uint32_t> _words;
};
typedef std::shared_ptr<DecoderWordTrace> DecoderWordTracePtr;
