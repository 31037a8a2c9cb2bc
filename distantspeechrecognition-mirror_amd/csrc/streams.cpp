// csrc/streams.cpp -- the stream/feature-operator API (include/dsr.h section 7).
//
// Mirrors FeatureStream<Type,item_type> (btk/stream/stream.h:36-75): reference-counted operators that
// hold their upstream(s), `next(frameX)` / `reset()` / `size()` / `name()` / `current()` / `isEnd()`,
// `frameX == -5` meaning "next", FrameResetX = -1, end of stream signalled as JITERATOR.
//
// The reference pulls one frame at a time through virtual calls; here an operator materialises its whole
// utterance on the device at the first next() after a reset() (every source of the path holds the full
// utterance in memory: SampleFeature::_samples, feature.cc:300-340) and then serves rows of its own
// host-side output buffer.  Compute always runs on the GPU (k_ops.hip / k_filterbank.hip / k_beamform.hip).
#include "common.h"
#include "ops.h"
#include "lattice.h"
#include <cmath>

using namespace dsr;

struct dsr_fb; struct dsr_bf; struct dsr_lpc; struct dsr_stft; struct dsr_prfb; struct dsr_zelinski; struct dsr_gmm; struct dsr_decoder;

struct dsr_stream {
  int refs = 1; std::string name; int size_ = 0; int type = DSR_T_FLOAT; int frameX = -1; bool endOfSamples = false;
  std::vector<dsr_stream*> ups;
  bool ready = false; int nFrames = 0;
  DevBuf<unsigned char> dev; std::vector<unsigned char> host;
  bool checkOrder = true;            // feature.cc operators throw jindex_error on out-of-order requests
  bool randomAccess = false;         // StorageFeature
  virtual ~dsr_stream() { for (size_t i = 0; i < ups.size(); i++) dsr_stream_release(ups[i]); }
  size_t itemsize() const { return type == DSR_T_CHAR ? 1 : type == DSR_T_SHORT ? 2 : type == DSR_T_FLOAT ? 4 : type == DSR_T_DOUBLE ? 8 : 16; }
  size_t rowBytes() const { return (size_t) size_ * itemsize(); }
  virtual void compute() = 0;        // fills dev (nFrames rows)
  virtual void reset() { frameX = -1; endOfSamples = false; ready = false; for (size_t i = 0; i < ups.size(); i++) ups[i]->reset(); }
  void add_up(dsr_stream* u) { dsr_stream_retain(u); ups.push_back(u); }
  void materialize() {
    if (ready) return;
    require_device();
    for (size_t i = 0; i < ups.size(); i++) ups[i]->materialize();
    compute();
    host.assign((size_t) nFrames * rowBytes() + 16, 0);
    if (nFrames > 0) { DSR_HIP(hipMemcpy(host.data(), dev.p, (size_t) nFrames * rowBytes(), hipMemcpyDeviceToHost)); }
    ready = true;
  }
  const void* row(int t) const { return host.data() + (size_t) t * rowBytes(); }
  virtual const void* next(int fx) {
    if (fx == frameX && frameX >= 0) return row(frameX);
    if (randomAccess && fx >= 0 && fx <= frameX) return row(fx);
    if (checkOrder && fx >= 0 && fx - 1 != frameX) throw Error(DSR_E_INDEX, "Problem in Feature %s: %d != %d", name.c_str(), fx - 1, frameX);
    materialize();
    if (frameX + 1 >= nFrames) { endOfSamples = true; throw Error(DSR_E_ITERATOR, "end of samples!"); }
    frameX++;
    return row(frameX);
  }
  template <class T> T* d() { return reinterpret_cast<T*>(dev.p); }
  void alloc(int T) { nFrames = T; dev.reserve((size_t) (T > 0 ? T : 1) * rowBytes()); }
};

namespace {

hipStream_t S0 = nullptr;

struct SampleSrc : dsr_stream {      // SampleFeature (feature.cc:222-689)
  int blockLen, shiftLen, padZeros; std::vector<float> samples; DevBuf<float> dx; int sampleRate = 16000, nChan = 1;
  void compute() override {
    const int n = (int) samples.size(); int T;
    if (padZeros) T = (n + shiftLen - 1) / shiftLen; else { long a = (long) n - blockLen; T = a > 0 ? (int) ((a + shiftLen - 1) / shiftLen) : 0; }
    dx.upload(samples.data(), samples.size() ? samples.size() : 0); if (!dx.p) dx.reserve(1);
    alloc(T); op_frames(dx.p, n, T, blockLen, shiftLen, d<float>(), S0);
  }
};
struct FrameSrc : dsr_stream {       // PyFeatureStream-like source: the caller hands over all frames (pyStream.h:44-130)
  std::vector<unsigned char> frames; int T = 0;
  // PyFeatureStream::reset() calls the Python object's reset() and starts a new iteration (pyStream.h:100-130).  A reset() that reaches this
  // source through a downstream operator's cascade marks the frames stale; the owner's refill callback (it calls the Python reset() and hands
  // the new frames over with dsr_frame_source_set_frames) runs before the next frame is served.
  int (*refill)(void*) = nullptr; void* refillUser = nullptr; bool stale = false, filling = false;
  void reset() override { dsr_stream::reset(); if (!filling) stale = true; }
  void compute() override {
    if (stale && refill) {
      filling = true; const int rc = refill(refillUser); filling = false; stale = false;
      if (rc != 0) throw Error(DSR_E_PYTHON, "the frame source's refill callback failed (%d)", rc);
    }
    stale = false;
    alloc(T); if (T > 0) DSR_HIP(hipMemcpy(dev.p, frames.data(), (size_t) T * rowBytes(), hipMemcpyHostToDevice));
  }
};
struct Preemph : dsr_stream { double mu; void compute() override { alloc(ups[0]->nFrames); op_preemph(ups[0]->d<float>(), nFrames, size_, mu, d<float>(), S0); } };
struct Hamming : dsr_stream {
  DevBuf<double> w;
  void compute() override {
    alloc(ups[0]->nFrames);
    if (ups[0]->type == DSR_T_SHORT) op_hamming_s(ups[0]->d<short>(), nFrames, size_, w.p, d<float>(), S0);
    else op_hamming_f(ups[0]->d<float>(), nFrames, size_, w.p, d<float>(), S0);
  }
};
struct HighPassOp : dsr_stream { int cut = 1; void compute() override { alloc(ups[0]->nFrames); op_highpass(ups[0]->d<double2>(), nFrames, size_, cut, d<double2>(), S0); } };   // highPassFilter (postfilter.cc:1222-1261)
struct FFTOp : dsr_stream { int L; DevBuf<double2> tw; void compute() override { alloc(ups[0]->nFrames); op_fft(ups[0]->d<float>(), nFrames, L, size_, tw.p, d<double2>(), S0); } };
struct PowerOp : dsr_stream { int fftLen; void compute() override { alloc(ups[0]->nFrames); op_power(ups[0]->d<double2>(), nFrames, fftLen, size_, d<double>(), S0); } };
struct VtlnOp : dsr_stream {
  DevBuf<int> s, c, o; DevBuf<double> coef, div; int rf = 0;
  void compute() override { alloc(ups[0]->nFrames); op_vtln(ups[0]->d<double>(), nFrames, size_, s.p, c.p, o.p, coef.p, div.p, rf, d<double>(), S0); }
};
struct MelOp : dsr_stream {
  DevBuf<int> s, c, o; DevBuf<float> coef; int inN;
  void compute() override { alloc(ups[0]->nFrames); op_mel(ups[0]->d<double>(), nFrames, inN, size_, s.p, c.p, o.p, coef.p, d<double>(), S0); }
};
struct LogOp : dsr_stream { double m, a; int sphinx; void compute() override { alloc(ups[0]->nFrames); op_log(ups[0]->d<double>(), (long) nFrames * size_, m, a, sphinx, d<float>(), S0); } };
struct GemvOp : dsr_stream {         // CepstralFeature and LinearTransformFeature
  DevBuf<float> A; std::vector<float> hA;
  void compute() override { alloc(ups[0]->nFrames); op_sgemv(ups[0]->d<float>(), nFrames, ups[0]->size_, size_, A.p, d<float>(), S0); }
};
struct StorageOp : dsr_stream {      // StorageFeature (feature.cc:2992-3085)
  void compute() override {
    if (ups[0]->nFrames > 100000) throw Error(DSR_E_DIMENSION, "Frame %d is greater than maximum number %d.", ups[0]->nFrames, 100000);
    alloc(ups[0]->nFrames); if (nFrames > 0) DSR_HIP(hipMemcpy(dev.p, ups[0]->dev.p, (size_t) nFrames * rowBytes(), hipMemcpyDeviceToDevice));
  }
};
struct LpcOp : dsr_stream {         // WarpMVDR/BurgMVDR/WarpLPC/BurgLPC features (lpc.h:86-195,262-331)
  dsr_lpc* plan = nullptr;
  ~LpcOp() override { if (plan) dsr_lpc_destroy(plan); }
  void compute() override {
    alloc(ups[0]->nFrames);
    if (nFrames > 0) { dsr_status s = dsr_lpc_run(plan, ups[0]->d<float>(), nFrames, d<double>(), S0); if (s) throw Error(s, "%s", dsr_last_error()); }
  }
};
struct CmnOp : dsr_stream {          // MeanSubtractionFeature(src, weight, devNormFactor, runon): ups[1] (optional) = the weight stream, element 0 of each frame
  int mode; double dnf;
  void compute() override {
    int T = ups[0]->nFrames;
    if (ups.size() > 1 && ups[1]->nFrames < T) T = ups[1]->nFrames;           // the weight stream ends first: so does the loop over the frames (feature.cc:2640-2644)
    alloc(T);
    op_cmn(ups[0]->d<float>(), nFrames, size_, mode, dnf, d<float>(), S0, ups.size() > 1 ? ups[1]->d<float>() : nullptr, ups.size() > 1 ? ups[1]->size_ : 0);
  }
};
struct AdjOp : dsr_stream {
  int delta;
  void compute() override { int T = ups[0]->nFrames; if (delta > 0 && T < delta) T = 0; alloc(T); op_adjacent(ups[0]->d<float>(), T, ups[0]->size_, delta, d<float>(), S0); }
};
struct AnalysisOp : dsr_stream {     // OverSampledDFTAnalysisBank
  dsr_fb* fb = nullptr; int M, D; DevBuf<float> x; DevBuf<float2> X; DevBuf<int> ns;
  ~AnalysisOp() override { if (fb) dsr_fb_destroy(fb); }
  void compute() override {
    // upstream delivers blocks of D samples (blockLen = shiftLen = D, padZeros): concatenate them again
    dsr_stream* u = ups[0]; const int nblk = u->nFrames; const int n = nblk * D;
    int T = dsr_fb_analysis_frames(fb, n); alloc(T);
    x.reserve(n > 0 ? n : 1); if (n > 0) DSR_HIP(hipMemcpy(x.p, u->dev.p, (size_t) n * sizeof(float), hipMemcpyDeviceToDevice));
    ns.upload(&n, 1);
    if (T > 0) {
      X.reserve((size_t) T * (M / 2 + 1));
      dsr_status s = dsr_fb_analysis(fb, x.p, ns.p, 1, 1, n > 0 ? n : 1, T, (float*) X.p, S0); if (s) throw Error(s, "%s", dsr_last_error());
      op_expand_bins(X.p, T, M / 2 + 1, M, d<double2>(), S0);
    }
  }
};
struct StftOp : dsr_stream {         // NormalFFTAnalysisBank (modulated.cc:121-257)
  dsr_stft* plan = nullptr; int M, D; DevBuf<float> x; DevBuf<float2> X; DevBuf<int> ns;
  ~StftOp() override { if (plan) dsr_stft_destroy(plan); }
  void compute() override {
    dsr_stream* u = ups[0]; const int nblk = u->nFrames; const int n = nblk * D;
    const int T = dsr_stft_frames(plan, n); alloc(T);
    x.reserve(n > 0 ? n : 1); if (n > 0) DSR_HIP(hipMemcpy(x.p, u->dev.p, (size_t) n * sizeof(float), hipMemcpyDeviceToDevice));
    ns.upload(&n, 1);
    X.reserve((size_t) T * M);
    dsr_status s = dsr_stft_analysis(plan, x.p, ns.p, 1, 1, n > 0 ? n : 1, T, (float*) X.p, S0); if (s) throw Error(s, "%s", dsr_last_error());
    op_expand_bins(X.p, T, M, M, d<double2>(), S0);          // all M bins are already there: widen to complex128
  }
};
struct SynthesisOp : dsr_stream {    // OverSampledDFTSynthesisBank
  dsr_fb* fb = nullptr; int M, D; DevBuf<float2> Y; DevBuf<int> nf;
  ~SynthesisOp() override { if (fb) dsr_fb_destroy(fb); }
  void compute() override {
    dsr_stream* u = ups[0]; const int Tin = u->nFrames; const int nb = dsr_fb_synthesis_blocks(fb, Tin); alloc(nb);
    if (nb <= 0) return;
    Y.reserve((size_t) Tin * (M / 2 + 1)); op_pack_hermitian(u->d<double2>(), Tin, M, Y.p, S0);      // frames need not be conjugate-symmetric (SubbandMMI + APAB)
    nf.upload(&Tin, 1);
    dsr_status s = dsr_fb_synthesis(fb, (const float*) Y.p, nf.p, 1, Tin, (int64_t) nb * D, d<float>(), S0); if (s) throw Error(s, "%s", dsr_last_error());
  }
};
struct PrAnalysisOp : dsr_stream {   // PerfectReconstructionFFTAnalysisBank (modulated.cc:686-818)
  dsr_prfb* fb = nullptr; int M2, D; DevBuf<float> x; DevBuf<float2> X; DevBuf<int> ns;
  ~PrAnalysisOp() override { if (fb) dsr_prfb_destroy(fb); }
  void compute() override {
    dsr_stream* u = ups[0]; const int nblk = u->nFrames; const int n = nblk * D;
    const int T = dsr_prfb_analysis_frames(fb, n); alloc(T);
    x.reserve(n > 0 ? n : 1); if (n > 0) DSR_HIP(hipMemcpy(x.p, u->dev.p, (size_t) n * sizeof(float), hipMemcpyDeviceToDevice));
    ns.upload(&n, 1); X.reserve((size_t) T * M2);
    dsr_status s = dsr_prfb_analysis(fb, x.p, ns.p, 1, 1, n > 0 ? n : 1, T, (float*) X.p, S0); if (s) throw Error(s, "%s", dsr_last_error());
    op_expand_bins(X.p, T, M2, M2, d<double2>(), S0);
  }
};
struct PrSynthesisOp : dsr_stream {  // PerfectReconstructionFFTSynthesisBank (modulated.cc:820-970)
  dsr_prfb* fb = nullptr; int M2, D; DevBuf<float2> Y; DevBuf<int> nf;
  ~PrSynthesisOp() override { if (fb) dsr_prfb_destroy(fb); }
  void compute() override {
    dsr_stream* u = ups[0]; const int Tin = u->nFrames; const int nb = dsr_prfb_synthesis_blocks(fb, Tin); alloc(nb);
    if (nb <= 0) return;
    Y.reserve((size_t) Tin * M2); op_pack_bins(u->d<double2>(), Tin, M2, M2, Y.p, S0);
    nf.upload(&Tin, 1);
    dsr_status s = dsr_prfb_synthesis(fb, (const float*) Y.p, nf.p, 1, Tin, (int64_t) nb * D, d<float>(), S0); if (s) throw Error(s, "%s", dsr_last_error());
  }
};
struct BfOp : dsr_stream {           // SubbandDS / SubbandGSC / SubbandMVDR as a stream
  dsr_bf* w; int M; DevBuf<float2> X, Y;
  void compute() override {
    const int C = (int) ups.size();
    if (C == 0 || C != dsr_bf_chan_n(w)) throw Error(DSR_E_DIMENSION, "Number of channels (%d) does not match the weights (%d)", C, dsr_bf_chan_n(w));
    int T = ups[0]->nFrames; for (int c = 1; c < C; c++) if (ups[c]->nFrames < T) T = ups[c]->nFrames;
    alloc(T); if (T <= 0) return;
    const int F = dsr_bf_bins(w); X.reserve((size_t) C * T * F); Y.reserve((size_t) T * F);       // M/2+1 unique bins, or all M with halfBandShift
    for (int c = 0; c < C; c++) op_pack_bins(ups[c]->d<double2>(), T, F, M, X.p + (size_t) c * T * F, S0);
    dsr_status s = dsr_bf_apply(w, (const float*) X.p, 1, T, (float*) Y.p, S0); if (s) throw Error(s, "%s", dsr_last_error());
    op_expand_bins(Y.p, T, F, M, d<double2>(), S0);
  }
};

struct MmiOp : BfOp {                // SubbandMMI as a stream (beamformer.cc:1973-2072); channels through dsr_subband_bf_set_channel
  dsr_mmi* mm = nullptr; DevBuf<int> nf;
  void compute() override {
    const int C = (int) ups.size();
    if (C == 0 || C != dsr_mmi_chan_n(mm)) throw Error(DSR_E_DIMENSION, "Number of channels (%d) does not match the weights (%d)", C, dsr_mmi_chan_n(mm));
    int T = ups[0]->nFrames; for (int c = 1; c < C; c++) if (ups[c]->nFrames < T) T = ups[c]->nFrames;
    alloc(T); if (T <= 0) return;
    const int F = dsr_mmi_bins(mm), Fo = dsr_mmi_out_bins(mm); X.reserve((size_t) C * T * F); Y.reserve((size_t) T * Fo);   // Fo = M: halfBandShift, or APAB's full frames
    for (int c = 0; c < C; c++) op_pack_bins(ups[c]->d<double2>(), T, F, M, X.p + (size_t) c * T * F, S0);
    nf.upload(&T, 1);
    dsr_status s = dsr_mmi_apply(mm, (const float*) X.p, nf.p, 1, T, (float*) Y.p, S0); if (s) throw Error(s, "%s", dsr_last_error());
    op_expand_bins(Y.p, T, Fo, M, d<double2>(), S0);
  }
};

struct OrthOp : dsr_stream {         // SubbandOrthogonalizer(beamformer, outChanX) (beamformer.cc:2817-2849): ups[0] = the SubbandMVDRGSC operator
  int outChanX = 0; DevBuf<float2> Z;
  void compute() override {
    BfOp* bf = dynamic_cast<BfOp*>(ups[0]); if (!bf) throw Error(DSR_E_PARAMETER, "SubbandOrthogonalizer needs a subband beamformer");
    const int T = bf->nFrames; alloc(T); if (T <= 0) return;
    if (outChanX <= 0) { DSR_HIP(hipMemcpyAsync(d<double2>(), bf->d<double2>(), sizeof(double2) * (size_t) T * size_, hipMemcpyDeviceToDevice, S0)); return; }
    const int F = bf->M / 2 + 1; Z.reserve((size_t) T * F);
    dsr_status s = dsr_bf_blocking_matrix_output(bf->w, (const float*) bf->X.p, 1, T, outChanX - 1, (float*) Z.p, S0); if (s) throw Error(s, "%s", dsr_last_error());
    op_orth_assemble(Z.p, bf->d<double2>(), T, F, bf->M, d<double2>(), S0);
  }
};
struct WpeOp : dsr_stream {          // SingleChannelWPEDereverberationFeature (dereverberation.cc:28-300)
  int M = 0, lowerN = 0, upperN = 0, iterationsN = 2; double loadDb = -20.0, bandWidth = 0.0, sampleRate = 16000.0; DevBuf<float2> Y, O; DevBuf<int> nf;
  void compute() override {
    const int T = ups[0]->nFrames; alloc(T); if (T <= 0) return;
    const int F = M / 2 + 1; Y.reserve((size_t) T * F); O.reserve((size_t) T * F);
    op_pack_bins(ups[0]->d<double2>(), T, F, M, Y.p, S0); nf.upload(&T, 1);
    dsr_status s = dsr_wpe_single((const float*) Y.p, nf.p, 1, T, M, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate, (float*) O.p, nullptr, S0);
    if (s) throw Error(s, "%s", dsr_last_error());
    op_expand_bins(O.p, T, F, M, d<double2>(), S0);
  }
};
struct WpeMultiOp : dsr_stream {     // MultiChannelWPEDereverberationFeature(source, channelX) (dereverberation.cc:281-602): ups = the source's channels
  int M = 0, lowerN = 0, upperN = 0, iterationsN = 2, channelX = 0, filterChan = -2; double loadDb = -20.0, bandWidth = 0.0, sampleRate = 16000.0;
  DevBuf<float2> Y, O; DevBuf<double2> G; DevBuf<int> nf;
  void compute() override {
    const int C = (int) ups.size();
    int T = ups[0]->nFrames; for (int c = 1; c < C; c++) if (ups[c]->nFrames < T) T = ups[c]->nFrames;      // _fillBuffer stops with the shortest channel (:397-412)
    alloc(T); if (T <= 0) return;
    const int F = M / 2 + 1, P = upperN - lowerN + 1; Y.reserve((size_t) C * T * F); O.reserve((size_t) C * T * F); G.reserve((size_t) C * F * C * P);
    for (int c = 0; c < C; c++) op_pack_bins(ups[c]->d<double2>(), T, F, M, Y.p + (size_t) c * T * F, S0);
    nf.upload(&T, 1);
    // all channels of a frame go through the filter of the channel that asked first (:381); on its own a feature asks first itself
    const int fc = filterChan == -2 ? channelX : filterChan;
    dsr_status s = dsr_wpe_multi((const float*) Y.p, nf.p, 1, C, T, M, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate, fc, (float*) O.p, (double*) G.p, S0);
    if (s) throw Error(s, "%s", dsr_last_error());
    op_expand_bins(O.p + (size_t) channelX * T * F, T, F, M, d<double2>(), S0);
  }
};
struct ZelinskiOp : dsr_stream {     // ZelinskiPostFilter (postfilter.cc:350-493): ups[0] = beamformer output, ups[1..] = the snapshot array's channels
  dsr_zelinski* plan = nullptr; int M = 0; double alpha = 0.6; int ptype = 2, minFrames = 0; std::vector<std::vector<double>> manifold; int chanSet = 0;
  int kind = 0; float threshold = 0.99f;                   // kind 1: McCowanPostFilter (the plan then also carries the noise coherence matrices)
  double minSV = 1e-8; int fbinX1 = 0;                     // kind 2: LefkimmiatisPostFilter
  void ensure_plan(int C) {
    if (plan) return;
    dsr_status s = kind == 2 ? dsr_lefkimmiatis_create(M, C, minSV, fbinX1, alpha, ptype, minFrames, threshold, &plan)
                 : kind ? dsr_mccowan_create(M, C, alpha, ptype, minFrames, threshold, &plan) : dsr_zelinski_create(M, C, alpha, ptype, minFrames, &plan);
    if (s) throw Error(s, "%s", dsr_last_error());
  }
  DevBuf<float2> X, Y, O; DevBuf<int> nf;
  ~ZelinskiOp() override { if (plan) dsr_zelinski_destroy(plan); }
  void compute() override {
    const int C = (int) ups.size() - 1;
    if (C < 1 || chanSet == 0) throw Error(DSR_E_ERROR, "set beamformer's weights");                     // postfilter.cc:447-450
    if (chanSet != C) throw Error(DSR_E_DIMENSION, "array manifold has %d channels, the snapshot array %d", chanSet, C);
    int T = ups[0]->nFrames; for (int c = 1; c <= C; c++) if (ups[c]->nFrames < T) T = ups[c]->nFrames;
    alloc(T); if (T <= 0) return;
    ensure_plan(C); dsr_status s;
    for (int f = 0; f <= M / 2; f++) if (!manifold[f].empty()) dsr_zelinski_set_manifold(plan, f, manifold[f].data());
    const int F = M / 2 + 1; X.reserve((size_t) C * T * F); Y.reserve((size_t) T * F); O.reserve((size_t) T * F);
    for (int c = 0; c < C; c++) op_pack_bins(ups[c + 1]->d<double2>(), T, F, M, X.p + (size_t) c * T * F, S0);
    op_pack_bins(ups[0]->d<double2>(), T, F, M, Y.p, S0);
    nf.upload(&T, 1);
    s = dsr_zelinski_apply(plan, (const float*) X.p, (const float*) Y.p, nf.p, 1, T, (float*) O.p, nullptr, S0); if (s) throw Error(s, "%s", dsr_last_error());
    op_expand_bins(O.p, T, F, M, d<double2>(), S0);
  }
};

template <class T> T* mk(const char* name, const char* dflt, int size, int type) { T* s = new T(); s->name = (name && *name) ? name : dflt; s->size_ = size; s->type = type; return s; }
dsr_stream* need(dsr_stream* s, int type, const char* what) {
  if (!s) throw Error(DSR_E_PARAMETER, "null upstream for %s", what);
  if (s->type != type) throw Error(DSR_E_TYPE, "%s needs an upstream of element type %d, got %d", what, type, s->type);
  return s;
}

}  // namespace

extern "C" {

void dsr_stream_retain(dsr_stream* s) { if (s) s->refs++; }
void dsr_stream_release(dsr_stream* s) { if (s && --s->refs == 0) delete s; }
int dsr_stream_size(const dsr_stream* s) { return s->size_; }
int dsr_stream_type(const dsr_stream* s) { return s->type; }
int dsr_stream_frameX(const dsr_stream* s) { return s->frameX; }
int dsr_stream_is_end(const dsr_stream* s) { return s->endOfSamples ? 1 : 0; }
const char* dsr_stream_name(const dsr_stream* s) { return s->name.c_str(); }

dsr_status dsr_stream_next(dsr_stream* s, int frameX, const void** data, size_t* n)
{ return guard([&] { if (!s || !data) throw Error(DSR_E_PARAMETER, "null argument"); *data = s->next(frameX); if (n) *n = (size_t) s->size_; }); }
dsr_status dsr_stream_current(dsr_stream* s, const void** data, size_t* n)
{
  return guard([&] {
    if (!s || !data) throw Error(DSR_E_PARAMETER, "null argument");
    if (s->frameX < 0) throw Error(DSR_E_CONSISTENCY, "Frame index (%d) < 0.", s->frameX);      // stream.h:44-46
    *data = s->next(s->frameX); if (n) *n = (size_t) s->size_;
  });
}
dsr_status dsr_stream_reset(dsr_stream* s) { return guard([&] { if (!s) throw Error(DSR_E_PARAMETER, "null argument"); s->reset(); }); }

dsr_status dsr_sample_feature_create(int blockLen, int shiftLen, int padZeros, const char* name, dsr_stream** out)
{
  return guard([&] {
    if (!out || blockLen < 1 || shiftLen < 1) throw Error(DSR_E_PARAMETER, "bad argument");
    SampleSrc* s = mk<SampleSrc>(name, "Sample", blockLen, DSR_T_FLOAT); s->blockLen = blockLen; s->shiftLen = shiftLen; s->padZeros = padZeros;
    s->checkOrder = true; *out = s;
  });
}
dsr_status dsr_sample_feature_set_samples(dsr_stream* s, const float* samples, size_t n, unsigned sampleRate)
{
  return guard([&] {
    SampleSrc* q = dynamic_cast<SampleSrc*>(s); if (!q || (!samples && n)) throw Error(DSR_E_PARAMETER, "not a SampleFeature");
    q->samples.assign(samples, samples + n); if (sampleRate) q->sampleRate = (int) sampleRate; q->reset();    // setSamples() resets (feature.cc:688)
  });
}
// SampleFeature::read(fn, format, samplerate, chX, chN, cfrom, to, outsamplerate, norm) (feature.cc:243-393).  The reference reads through libsndfile's
// sf_readf_float; here: RIFF/WAVE PCM of 8, 16, 24 or 32 bits (format tag 1, or the extensible tag carrying PCM), parsed by hand.  norm == 0 keeps the
// file's integer scale (SFC_SET_NORM_FLOAT off), otherwise samples are normalised to [-1, 1) and, for norm != 1, multiplied by norm.  The error branches
// are the reference's: a file that cannot be opened or parsed and an empty sample range are jio errors, chX == 0 and chX out of range jconsistency
// errors (in the reference's order: the range is checked before the channel).  Sample-rate conversion (outsamplerate != the file's rate; SRCONV builds
// only) is refused.  format / samplerate / chN only steer libsndfile's RAW reader and are accepted for the signature.  *nread = frames read.
dsr_status dsr_sample_feature_read(dsr_stream* s, const char* fn, int format, int samplerate, int chX, int chN, int cfrom, int to, int outsamplerate,
                                   float norm, int* nread)
{
  (void) format; (void) samplerate; (void) chN;
  return guard([&] {
    SampleSrc* q = dynamic_cast<SampleSrc*>(s); if (!q || !fn) throw Error(DSR_E_PARAMETER, "not a SampleFeature");
    FILE* fp = fopen(fn, "rb");
    if (!fp) throw Error(DSR_E_IO, "Could not open file %s.", fn);
    std::vector<unsigned char> raw; int nch = 0, bits = 0, rate = 0; long frames = 0; int sw = 0;
    try {
      auto rd = [&](void* p, size_t n) { if (fread(p, 1, n, fp) != n) throw Error(DSR_E_IO, "Could not open file %s.", fn); };
      auto u32 = [&]() { unsigned char b[4]; rd(b, 4); return (unsigned) b[0] | (unsigned) b[1] << 8 | (unsigned) b[2] << 16 | (unsigned) b[3] << 24; };
      auto u16 = [&]() { unsigned char b[2]; rd(b, 2); return (unsigned) (b[0] | b[1] << 8); };
      char id[4]; rd(id, 4); if (memcmp(id, "RIFF", 4)) throw Error(DSR_E_IO, "Could not open file %s.", fn);
      (void) u32(); rd(id, 4); if (memcmp(id, "WAVE", 4)) throw Error(DSR_E_IO, "Could not open file %s.", fn);
      bool haveFmt = false, haveData = false; long dataBytes = 0;
      while (!haveData) {
        if (fread(id, 1, 4, fp) != 4) break;
        const unsigned len = u32();
        if (!memcmp(id, "fmt ", 4)) {
          if (len < 16) throw Error(DSR_E_IO, "Could not open file %s.", fn);
          const unsigned tag = u16(); nch = (int) u16(); rate = (int) u32(); (void) u32(); (void) u16(); bits = (int) u16();
          if (tag != 1 && tag != 0xFFFE) throw Error(DSR_E_IO, "Could not open file %s.", fn);
          if (len > 16) fseek(fp, (long) (len - 16 + (len & 1)), SEEK_CUR);
          haveFmt = true;
        } else if (!memcmp(id, "data", 4)) {
          if (!haveFmt) throw Error(DSR_E_IO, "Could not open file %s.", fn);
          dataBytes = (long) len; haveData = true;
        } else fseek(fp, (long) (len + (len & 1)), SEEK_CUR);
      }
      if (!haveData || nch < 1) throw Error(DSR_E_IO, "Could not open file %s.", fn);
      sw = (bits + 7) / 8;
      if (sw < 1 || sw > 4) throw Error(DSR_E_IO, "sndfile error: unsupported sample width %d.", sw);
      frames = dataBytes / ((long) sw * nch);
      if (outsamplerate == -1) outsamplerate = rate;
      if (to < 0 || to >= frames) to = (int) frames - 1;
      if (cfrom < 0) cfrom = 0;
      if (cfrom > to || cfrom > frames) throw Error(DSR_E_IO, "Cannot load samples from %d to %d.", cfrom, to);
      const long n = (long) to - cfrom + 1;
      fseek(fp, (long) cfrom * sw * nch, SEEK_CUR);
      raw.resize((size_t) n * sw * nch);
      const size_t got = fread(raw.data(), 1, raw.size(), fp); raw.resize(got - got % ((size_t) sw * nch));
    } catch (...) { fclose(fp); throw; }
    fclose(fp);
    if (chX > nch || chX < 1) {
      if (chX == 0) throw Error(DSR_E_CONSISTENCY, "Multi-channel read is not yet supported.");
      throw Error(DSR_E_CONSISTENCY, "Selected channel out of range of available channels.");
    }
    const size_t nfr = raw.size() / ((size_t) sw * nch);
    std::vector<float> x(nfr);
    for (size_t i = 0; i < nfr; i++) {
      const unsigned char* b = raw.data() + (i * nch + (size_t) (chX - 1)) * sw; long long v;
      if (sw == 1) v = (long long) b[0] - 128;
      else if (sw == 2) v = (short) (b[0] | b[1] << 8);
      else if (sw == 3) { int t = b[0] | b[1] << 8 | b[2] << 16; if (t >= 1 << 23) t -= 1 << 24; v = t; }
      else v = (int) ((unsigned) b[0] | (unsigned) b[1] << 8 | (unsigned) b[2] << 16 | (unsigned) b[3] << 24);
      double d = (double) v;
      if (norm != 0.0f) d = d / (double) (1LL << (8 * sw - 1));                            // libsndfile's float normalisation
      x[i] = (float) d;
    }
    if (rate != outsamplerate) throw Error(DSR_E_ERROR, "sample rate conversion (%d -> %d) is not supported", rate, outsamplerate);
    if (norm != 1.0f && norm != 0.0f) for (size_t i = 0; i < nfr; i++) x[i] *= norm;
    q->samples.swap(x); q->sampleRate = rate; q->nChan = nch; q->reset();                  // _cur = 0; reset() (:386-388)
    if (nread) *nread = (int) nfr;
  });
}
int dsr_sample_feature_sample_rate(const dsr_stream* s) { const SampleSrc* q = dynamic_cast<const SampleSrc*>(s); return q ? q->sampleRate : 0; }

dsr_status dsr_frame_source_create(int type, int size, const char* name, dsr_stream** out)
{
  return guard([&] {
    if (!out || size < 1 || type < 0 || type > DSR_T_COMPLEX) throw Error(DSR_E_PARAMETER, "bad argument");
    FrameSrc* s = mk<FrameSrc>(name, "PyFeatureStream", size, type); s->checkOrder = false; *out = s;
  });
}
dsr_status dsr_frame_source_set_frames(dsr_stream* s, const void* data, size_t nframes)
{
  return guard([&] {
    FrameSrc* q = dynamic_cast<FrameSrc*>(s); if (!q || (!data && nframes)) throw Error(DSR_E_PARAMETER, "not a frame source");
    q->frames.assign((const unsigned char*) data, (const unsigned char*) data + nframes * q->rowBytes()); q->T = (int) nframes;
    const bool f = q->filling; q->filling = true; q->reset(); q->filling = f; q->stale = false;          // fresh frames: nothing to refill
  });
}
dsr_status dsr_frame_source_set_refill(dsr_stream* s, int (*refill)(void*), void* user)
{
  return guard([&] {
    FrameSrc* q = dynamic_cast<FrameSrc*>(s); if (!q) throw Error(DSR_E_PARAMETER, "not a frame source");
    q->refill = refill; q->refillUser = user;
  });
}

dsr_status dsr_analysis_bank_create(dsr_stream* samp, const double* prototype, int M, int m, int r, int dct, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(samp, DSR_T_FLOAT, "OverSampledDFTAnalysisBank"); if (!out || !prototype) throw Error(DSR_E_PARAMETER, "null argument");
    const int D = M >> r;
    if (samp->size_ != D) throw Error(DSR_E_DIMENSION, "Input block length (%d) != _D (%d)", samp->size_, D);      // modulated.cc:373-374
    AnalysisOp* s = mk<AnalysisOp>(name, "OverSampledDFTAnalysisBank", M, DSR_T_COMPLEX); s->M = M; s->D = D; s->checkOrder = false;
    dsr_status st = dsr_fb_create(prototype, M, m, r, 0, dct, 1, &s->fb); if (st) { delete s; throw Error(st, "%s", dsr_last_error()); }
    s->add_up(samp); *out = s;
  });
}
dsr_status dsr_pr_analysis_bank_create(dsr_stream* samp, const double* prototype, int M, int m, int r, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(samp, DSR_T_FLOAT, "PerfectReconstructionFFTAnalysisBank"); if (!out || !prototype) throw Error(DSR_E_PARAMETER, "null argument");
    const int D = M >> r;
    if (samp->size_ != D) throw Error(DSR_E_DIMENSION, "Input block length (%d) != _D (%d)", samp->size_, D);
    PrAnalysisOp* s = mk<PrAnalysisOp>(name, "PerfectReconstructionFFTAnalysisBank", 2 * M, DSR_T_COMPLEX); s->M2 = 2 * M; s->D = D; s->checkOrder = false;
    dsr_status st = dsr_prfb_create(prototype, M, m, r, &s->fb); if (st) { delete s; throw Error(st, "%s", dsr_last_error()); }
    s->add_up(samp); *out = s;
  });
}
dsr_status dsr_pr_synthesis_bank_create(dsr_stream* samp, const double* prototype, int M, int m, int r, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(samp, DSR_T_COMPLEX, "PerfectReconstructionFFTSynthesisBank"); if (!out || !prototype) throw Error(DSR_E_PARAMETER, "null argument");
    if (samp->size_ != 2 * M) throw Error(DSR_E_DIMENSION, "Input size (%d) != 2M (%d)", samp->size_, 2 * M);
    PrSynthesisOp* s = mk<PrSynthesisOp>(name, "PerfectReconstructionFFTSynthesisBank", M >> r, DSR_T_FLOAT); s->M2 = 2 * M; s->D = M >> r; s->checkOrder = false;
    dsr_status st = dsr_prfb_create(prototype, M, m, r, &s->fb); if (st) { delete s; throw Error(st, "%s", dsr_last_error()); }
    s->add_up(samp); *out = s;
  });
}
dsr_status dsr_normal_fft_bank_create(dsr_stream* samp, int M, int r, int windowType, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(samp, DSR_T_FLOAT, "NormalFFTAnalysisBank"); if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    const int D = M >> r;
    if (samp->size_ != D) throw Error(DSR_E_DIMENSION, "Input block length (%d) != _D (%d)", samp->size_, D);      // modulated.cc:138-139
    StftOp* s = mk<StftOp>(name, "NormalFFTAnalysisBank", M, DSR_T_COMPLEX); s->M = M; s->D = D; s->checkOrder = false;
    dsr_status st = dsr_stft_create(M, r, windowType, &s->plan); if (st) { delete s; throw Error(st, "%s", dsr_last_error()); }
    s->add_up(samp); *out = s;
  });
}
dsr_status dsr_synthesis_bank_create(dsr_stream* samp, const double* prototype, int M, int m, int r, int dct, int gain, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(samp, DSR_T_COMPLEX, "OverSampledDFTSynthesisBank"); if (!out || !prototype) throw Error(DSR_E_PARAMETER, "null argument");
    if (samp->size_ != M) throw Error(DSR_E_DIMENSION, "Input size (%d) != M (%d)", samp->size_, M);
    SynthesisOp* s = mk<SynthesisOp>(name, "OverSampledDFTSynthesisBank", M >> r, DSR_T_FLOAT); s->M = M; s->D = M >> r; s->checkOrder = false;
    dsr_status st = dsr_fb_create(prototype, M, m, r, 1, dct, gain, &s->fb); if (st) { delete s; throw Error(st, "%s", dsr_last_error()); }
    s->add_up(samp); *out = s;
  });
}
dsr_status dsr_wpe_single_stream_create(dsr_stream* samples, int lowerN, int upperN, int iterationsN, double loadDb, double bandWidth, double sampleRate,
                                        const char* name, dsr_stream** out)
{
  return guard([&] {
    need(samples, DSR_T_COMPLEX, "SingleChannelWPEDereverberationFeature"); if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (upperN < lowerN) throw Error(DSR_E_PARAMETER, "bad prediction range [%d, %d]", lowerN, upperN);
    if (bandWidth > sampleRate / 2.0) throw Error(DSR_E_DIMENSION, "Bandwidth is greater than the Nyquist rate.");
    WpeOp* s = mk<WpeOp>(name, "SingleChannelWPEDereverberationFeature", samples->size_, DSR_T_COMPLEX);
    s->M = samples->size_; s->lowerN = lowerN; s->upperN = upperN; s->iterationsN = iterationsN; s->loadDb = loadDb; s->bandWidth = bandWidth; s->sampleRate = sampleRate;
    s->add_up(samples); *out = s;
  });
}
dsr_status dsr_wpe_multi_feature_create(dsr_stream* const* channels, int channelsN, int channelX, int lowerN, int upperN, int iterationsN, double loadDb,
                                        double bandWidth, double sampleRate, const char* name, dsr_stream** out)
{
  return guard([&] {
    if (!channels || !out || channelsN < 1) throw Error(DSR_E_PARAMETER, "null argument");
    if (channelX < 0 || channelX >= channelsN) throw Error(DSR_E_INDEX, "channel %d of %d", channelX, channelsN);
    for (int c = 0; c < channelsN; c++) { need(channels[c], DSR_T_COMPLEX, "MultiChannelWPEDereverberation"); if (channels[c]->size_ != channels[0]->size_) throw Error(DSR_E_DIMENSION, "channel %d has %d subbands, channel 0 %d", c, channels[c]->size_, channels[0]->size_); }
    if (upperN < lowerN) throw Error(DSR_E_PARAMETER, "bad prediction range [%d, %d]", lowerN, upperN);
    if (bandWidth > sampleRate / 2.0) throw Error(DSR_E_DIMENSION, "Bandwidth is greater than the Nyquist rate.");
    WpeMultiOp* s = mk<WpeMultiOp>(name, "MultiChannelWPEDereverberationFeature", channels[0]->size_, DSR_T_COMPLEX);
    s->M = channels[0]->size_; s->channelX = channelX; s->lowerN = lowerN; s->upperN = upperN; s->iterationsN = iterationsN; s->loadDb = loadDb; s->bandWidth = bandWidth; s->sampleRate = sampleRate;
    for (int c = 0; c < channelsN; c++) s->add_up(channels[c]);
    s->checkOrder = true; *out = s;                      // getOutput: jindex_error on out-of-order requests (:371-372)
  });
}
dsr_status dsr_wpe_multi_feature_set_filter_channel(dsr_stream* feature, int filterChan)
{
  return guard([&] {
    WpeMultiOp* q = dynamic_cast<WpeMultiOp*>(feature); if (!q) throw Error(DSR_E_PARAMETER, "not a MultiChannelWPEDereverberationFeature");
    if (filterChan >= (int) q->ups.size()) throw Error(DSR_E_INDEX, "filter channel %d of %d", filterChan, (int) q->ups.size());
    q->filterChan = filterChan < 0 ? -1 : filterChan; q->ready = false;
  });
}
dsr_status dsr_zelinski_stream_create(dsr_stream* output, int fftLen, double alpha, int type, int minFrames, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(output, DSR_T_COMPLEX, "ZelinskiPostFilter"); if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (output->size_ != fftLen) throw Error(DSR_E_DIMENSION, "Input block length (%d) != fftLen (%d)", output->size_, fftLen);      // postfilter.cc:359-362
    ZelinskiOp* s = mk<ZelinskiOp>(name, "ZelinskPostFilter", fftLen, DSR_T_COMPLEX); s->M = fftLen; s->alpha = alpha; s->ptype = type; s->minFrames = minFrames;
    s->manifold.assign((size_t) fftLen / 2 + 1, std::vector<double>()); s->checkOrder = false;
    s->add_up(output); *out = s;
  });
}
dsr_status dsr_mccowan_stream_create(dsr_stream* output, int fftLen, double alpha, int type, int minFrames, float threshold, const char* name, dsr_stream** out)
{
  const dsr_status s0 = dsr_zelinski_stream_create(output, fftLen, alpha, type, minFrames, (name && *name) ? name : "McCowanPostFilterPtr", out);
  if (s0 != DSR_OK) return s0;
  ZelinskiOp* q = static_cast<ZelinskiOp*>(*out); q->kind = 1; q->threshold = threshold;
  return DSR_OK;
}
dsr_status dsr_lefkimmiatis_stream_create(dsr_stream* output, int fftLen, double minSV, int fbinX1, double alpha, int type, int minFrames, float threshold,
                                          const char* name, dsr_stream** out)
{
  const dsr_status s0 = dsr_zelinski_stream_create(output, fftLen, alpha, type, minFrames, (name && *name) ? name : "LefkimmiatisPostFilte", out);
  if (s0 != DSR_OK) return s0;
  ZelinskiOp* q = static_cast<ZelinskiOp*>(*out); q->kind = 2; q->threshold = threshold; q->minSV = minSV; q->fbinX1 = fbinX1;
  return DSR_OK;
}
// noise coherence setters of McCowanPostFilter (postfilter.cc:546-682); chanN fixes the array size at the first call
dsr_status dsr_mccowan_stream_set_noise(dsr_stream* pf, int what, int fbinX, const double* data, int chanN, double a, double b)
{
  return guard([&] {
    ZelinskiOp* q = dynamic_cast<ZelinskiOp*>(pf); if (!q || q->kind < 1) throw Error(DSR_E_PARAMETER, "not a McCowan post-filter");
    q->ensure_plan(chanN); dsr_status s = DSR_OK;
    switch (what) {
    case 0: s = dsr_mccowan_set_noise_matrix(q->plan, fbinX, data); break;
    case 1: s = dsr_mccowan_set_diffuse_noise_model(q->plan, data, a, b); break;
    case 2: s = dsr_mccowan_diagonal_loading(q->plan, fbinX, (float) a); break;
    case 3: s = dsr_mccowan_divide_nondiagonal(q->plan, (float) a); break;
    default: throw Error(DSR_E_PARAMETER, "bad selector %d", what);
    }
    if (s) throw Error(s, "%s", dsr_last_error());
    q->ready = false;
  });
}
dsr_status dsr_zelinski_stream_set_channel(dsr_stream* pf, dsr_stream* chan)
{
  return guard([&] {
    ZelinskiOp* q = dynamic_cast<ZelinskiOp*>(pf); if (!q) throw Error(DSR_E_PARAMETER, "not a Zelinski post-filter");
    need(chan, DSR_T_COMPLEX, "ZelinskiPostFilter channel"); if (chan->size_ != q->M) throw Error(DSR_E_DIMENSION, "channel size %d != fftLen %d", chan->size_, q->M);
    q->add_up(chan); q->ready = false;
  });
}
dsr_status dsr_zelinski_stream_set_manifold(dsr_stream* pf, int fbinX, const double* vec, int chanN)
{
  return guard([&] {
    ZelinskiOp* q = dynamic_cast<ZelinskiOp*>(pf); if (!q || !vec) throw Error(DSR_E_PARAMETER, "not a Zelinski post-filter");
    if (fbinX < 0 || fbinX >= q->M) throw Error(DSR_E_DIMENSION, "fbinX %d must be less than %d", fbinX, q->M);
    if (fbinX <= q->M / 2) q->manifold[fbinX].assign(vec, vec + 2 * (size_t) chanN);
    q->chanSet = chanN; q->ready = false;
  });
}
dsr_status dsr_subband_bf_create(dsr_bf* weights, const char* name, dsr_stream** out)
{
  return guard([&] {
    if (!weights || !out) throw Error(DSR_E_PARAMETER, "null argument");
    BfOp* s = mk<BfOp>(name, "SubbandBeamformer", dsr_bf_fft_len(weights), DSR_T_COMPLEX); s->w = weights; s->M = dsr_bf_fft_len(weights); s->checkOrder = false; *out = s;
  });
}
dsr_status dsr_subband_mmi_stream_create(dsr_mmi* weights, int fftLen, const char* name, dsr_stream** out)
{
  return guard([&] {
    if (!weights || !out) throw Error(DSR_E_PARAMETER, "null argument");
    MmiOp* s = mk<MmiOp>(name, "SubbandMMI", fftLen, DSR_T_COMPLEX); s->w = nullptr; s->mm = weights; s->M = fftLen; s->checkOrder = false; *out = s;
  });
}
dsr_status dsr_subband_orthogonalizer_create(dsr_stream* beamformer, int outChanX, const char* name, dsr_stream** out)
{
  return guard([&] {
    BfOp* q = dynamic_cast<BfOp*>(beamformer); if (!q || !q->w || !out) throw Error(DSR_E_PARAMETER, "not a subband beamformer");
    OrthOp* s = mk<OrthOp>(name, "SubbandOrthogonalizer", q->M, DSR_T_COMPLEX); s->outChanX = outChanX; s->checkOrder = false;
    s->add_up(beamformer); *out = s;
  });
}
dsr_status dsr_subband_bf_set_channel(dsr_stream* bf, dsr_stream* chan)
{
  return guard([&] {
    BfOp* q = dynamic_cast<BfOp*>(bf); if (!q) throw Error(DSR_E_PARAMETER, "not a subband beamformer");
    need(chan, DSR_T_COMPLEX, "setChannel"); if (chan->size_ != q->M) throw Error(DSR_E_DIMENSION, "channel size %d != fftLen %d", chan->size_, q->M);
    q->add_up(chan); q->ready = false;
  });
}
dsr_status dsr_preemphasis_create(dsr_stream* samp, double mu, const char* name, dsr_stream** out)
{ return guard([&] { need(samp, DSR_T_FLOAT, "PreemphasisFeature"); Preemph* s = mk<Preemph>(name, "Preemphasis", samp->size_, DSR_T_FLOAT); s->mu = mu; s->add_up(samp); *out = s; }); }
dsr_status dsr_hamming_create(dsr_stream* samp, const char* name, dsr_stream** out)
{
  return guard([&] {
    if (!samp || (samp->type != DSR_T_FLOAT && samp->type != DSR_T_SHORT)) throw Error(DSR_E_TYPE, "HammingFeature needs a float or short stream");
    Hamming* s = mk<Hamming>(name, "Hamming", samp->size_, DSR_T_FLOAT);
    std::vector<double> w(samp->size_); const double temp = 2. * M_PI / (double) (samp->size_ - 1);
    for (int i = 0; i < samp->size_; i++) w[i] = 0.54 - 0.46 * cos(temp * i);
    require_device(); s->w.upload(w); s->add_up(samp); *out = s;
  });
}
dsr_status dsr_highpass_filter_create(dsr_stream* output, float cutOffFreq, int sampleRate, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(output, DSR_T_COMPLEX, "highPassFilter"); if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (sampleRate <= 0) throw Error(DSR_E_PARAMETER, "sample rate %d", sampleRate);
    const unsigned cut = (unsigned) ((float) output->size_ * cutOffFreq / (float) sampleRate);                 // postfilter.cc:1228
    if (cut < 1 || cut > (unsigned) output->size_ / 2) throw Error(DSR_E_INDEX, "highPassFilter: the cut-off bin %u must lie in 1..%d (the reference writes outside its vector otherwise)", cut, output->size_ / 2);
    HighPassOp* s = mk<HighPassOp>(name, "highPassFilter", output->size_, DSR_T_COMPLEX); s->cut = (int) cut; s->checkOrder = false;
    s->add_up(output); *out = s;
  });
}
dsr_status dsr_fft_create(dsr_stream* samp, int fftLen, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(samp, DSR_T_FLOAT, "FFTFeature");
    if (!is_pow2((unsigned) fftLen) || fftLen < 2 || fftLen > 8192 || samp->size_ > fftLen) throw Error(DSR_E_DIMENSION, "fftLen=%d must be a power of two >= the window length %d", fftLen, samp->size_);
    FFTOp* s = mk<FFTOp>(name, "FFT", fftLen, DSR_T_COMPLEX); s->L = samp->size_;
    std::vector<double2> tw(fftLen); for (int k = 0; k < fftLen; k++) { const double a = 2.0 * M_PI * k / fftLen; tw[k] = make_double2(cos(a), sin(a)); }
    require_device(); s->tw.upload(tw); s->add_up(samp); *out = s;
  });
}
dsr_status dsr_spectral_power_create(dsr_stream* fft, int powN, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(fft, DSR_T_COMPLEX, "SpectralPowerFeature"); const int sz = powN == 0 ? fft->size_ : powN;
    if (sz != fft->size_ && sz != fft->size_ / 2 + 1) throw Error(DSR_E_CONSISTENCY, "Number of power coefficients %d does not match FFT length %d.", sz, fft->size_);
    PowerOp* s = mk<PowerOp>(name, "Power", sz, DSR_T_DOUBLE); s->fftLen = fft->size_; s->add_up(fft); *out = s;
  });
}
dsr_status dsr_vtln_create(dsr_stream* pow, int coeffN, double ratio, double edge, int version, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(pow, DSR_T_DOUBLE, "VTLNFeature"); const int N = coeffN == 0 ? pow->size_ : coeffN;
    if (N != pow->size_) throw Error(DSR_E_DIMENSION, "VTLN size %d != input size %d", N, pow->size_);
    if (version != 1 && version != 2) throw Error(DSR_E_PARAMETER, "unknown version number (%d)", version);
    VtlnOp* s = mk<VtlnOp>(name, "VTLN", N, DSR_T_DOUBLE); SparseRowsD r; build_vtln_rows(N, ratio, edge, version, r);
    require_device(); s->s.upload(r.start); s->c.upload(r.count); s->o.upload(r.off); s->coef.upload(r.coef); s->div.upload(r.div); s->rf = r.roundFloat;
    s->add_up(pow); *out = s;
  });
}
dsr_status dsr_mel_create(dsr_stream* mag, int powN, float rate, float low, float up, int filterN, int version, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(mag, DSR_T_DOUBLE, "MelFeature"); const int P = powN == 0 ? mag->size_ : powN;
    MelOp* s = mk<MelOp>(name, "MelFFT", filterN, DSR_T_DOUBLE); SparseRowsF r; build_mel_rows(P, rate, low, up, filterN, version, r);
    if (mag->size_ < r.nReq) { delete s; throw Error(DSR_E_CONSISTENCY, "Matrix columns differ: %d and %d.", mag->size_, r.nReq); }
    for (size_t i = 0; i < r.start.size(); i++) if (r.start[i] + r.count[i] > mag->size_) { delete s; throw Error(DSR_E_CONSISTENCY, "mel filter %zu reads past the input", i); }
    require_device(); s->s.upload(r.start); s->c.upload(r.count); s->o.upload(r.off); s->coef.upload(r.coef); s->inN = mag->size_;
    s->add_up(mag); *out = s;
  });
}
dsr_status dsr_log_create(dsr_stream* mel, double m, double a, int sphinx, const char* name, dsr_stream** out)
{ return guard([&] { need(mel, DSR_T_DOUBLE, "LogFeature"); LogOp* s = mk<LogOp>(name, "LogMel", mel->size_, DSR_T_FLOAT); s->m = m; s->a = a; s->sphinx = sphinx; s->add_up(mel); *out = s; }); }
dsr_status dsr_cepstral_create(dsr_stream* mel, int ncep, int type, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(mel, DSR_T_FLOAT, "CepstralFeature");
    GemvOp* s = mk<GemvOp>(name, "Cepstral", ncep, DSR_T_FLOAT); build_dct(ncep, mel->size_, type, s->hA);
    require_device(); s->A.upload(s->hA); s->add_up(mel); *out = s;
  });
}
dsr_status dsr_lpc_feature_create(dsr_stream* src, int order, int correlate, float warp, int method, int kind, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(src, DSR_T_FLOAT, kind ? "LPCFeature" : "MVDRFeature");
    dsr_lpc* plan = nullptr;
    dsr_status s0 = dsr_lpc_create(src->size_, order, correlate, warp, method, kind, &plan); if (s0) throw Error(s0, "%s", dsr_last_error());
    LpcOp* s = mk<LpcOp>(name, kind ? "LPC" : "MVDR", src->size_ / 2 + 1, DSR_T_DOUBLE); s->plan = plan; s->add_up(src); *out = s;
  });
}
dsr_status dsr_storage_create(dsr_stream* src, const char* name, dsr_stream** out)
{ return guard([&] { need(src, DSR_T_FLOAT, "StorageFeature"); StorageOp* s = mk<StorageOp>(name, "Storage", src->size_, DSR_T_FLOAT); s->randomAccess = true; s->checkOrder = false; s->add_up(src); *out = s; }); }
dsr_status dsr_mean_subtraction_create(dsr_stream* src, double dnf, int runon, const char* name, dsr_stream** out)
{ return guard([&] { need(src, DSR_T_FLOAT, "MeanSubtractionFeature"); CmnOp* s = mk<CmnOp>(name, "Mean Subtraction", src->size_, DSR_T_FLOAT); s->mode = runon ? 2 : 1; s->dnf = dnf; s->add_up(src); *out = s; }); }
dsr_status dsr_mean_subtraction_set_weight(dsr_stream* cmn, dsr_stream* weight)
{
  return guard([&] {
    CmnOp* q = dynamic_cast<CmnOp*>(cmn); if (!q) throw Error(DSR_E_PARAMETER, "not a MeanSubtractionFeature");
    need(weight, DSR_T_FLOAT, "MeanSubtractionFeature weight"); if (q->ups.size() > 1) throw Error(DSR_E_CONSISTENCY, "the weight stream is already set");
    q->add_up(weight); q->ready = false;
  });
}
dsr_status dsr_adjacent_create(dsr_stream* single, int delta, const char* name, dsr_stream** out)
{ return guard([&] { need(single, DSR_T_FLOAT, "AdjacentFeature"); AdjOp* s = mk<AdjOp>(name, "Adjacent", (2 * delta + 1) * single->size_, DSR_T_FLOAT); s->delta = delta; s->add_up(single); *out = s; }); }
dsr_status dsr_linear_transform_create(dsr_stream* src, int sz, const char* name, dsr_stream** out)
{
  return guard([&] {
    need(src, DSR_T_FLOAT, "LinearTransformFeature"); if (sz < 1) throw Error(DSR_E_DIMENSION, "bad output size %d", sz);
    GemvOp* s = mk<GemvOp>(name, "Transform", sz, DSR_T_FLOAT); s->hA.assign((size_t) sz * src->size_, 0.0f);     // calloc'd matrix (feature.cc:2936)
    require_device(); s->A.upload(s->hA); s->add_up(src); *out = s;
  });
}
dsr_status dsr_linear_transform_set(dsr_stream* s, const float* matrix)
{
  return guard([&] {
    GemvOp* q = dynamic_cast<GemvOp*>(s); if (!q || !matrix) throw Error(DSR_E_PARAMETER, "not a linear transform");
    q->hA.assign(matrix, matrix + q->hA.size()); q->A.upload(q->hA); q->ready = false;
  });
}

// LinearTransformFeature::load(fileName, old) (feature.cc:2972-2976): gsl_matrix_float_load into the operator's (size x srcSize) matrix, then the
// crop to (size x srcSize) -- which only ever shrinks (gslmatrix.cc:6-15), so a Janus file must carry exactly that shape.
dsr_status dsr_linear_transform_load(dsr_stream* s, const char* fileName, int old)
{
  return guard([&] {
    GemvOp* q = dynamic_cast<GemvOp*>(s); if (!q || !fileName) throw Error(DSR_E_PARAMETER, "not a linear transform");
    const int rows = q->size_, cols = q->ups[0]->size_;
    std::vector<float> m((size_t) rows * cols, 0.0f); int r2 = 0, c2 = 0;
    const dsr_status st = dsr_fmat_load(fileName, old, rows, cols, m.data(), &r2, &c2); if (st) throw Error(st, "%s", dsr_last_error());
    if (r2 < rows) throw Error(DSR_E_DIMENSION, "Cannot resize from %d to %d", r2, rows);           // the crop back to (size x srcSize)
    if (c2 < cols) throw Error(DSR_E_DIMENSION, "Cannot resize from %d to %d", c2, cols);
    q->hA = m; q->A.upload(q->hA); q->ready = false;
  });
}
// StorageFeature::write(fileName, plainText) / read(fileName) (feature.cc:3025-3067), quirks kept: the count that is written is _frameX -- the
// index of the last frame, one less than the number of frames that follow it; read() takes that number for _frameX and reads that many frames,
// i.e. one fewer than the file holds.  Binary: big-endian ints (write_int) and native-endian float blocks (gsl_vector_float_fwrite).
dsr_status dsr_storage_write(dsr_stream* s, const char* fileName, int plainText)
{
  return guard([&] {
    StorageOp* q = dynamic_cast<StorageOp*>(s); if (!q || !fileName) throw Error(DSR_E_PARAMETER, "not a StorageFeature");
    if (q->frameX <= 0) throw Error(DSR_E_IO, "Frame count must be > 0.\n");
    FILE* fp = fopen(fileName, "w"); if (!fp) throw Error(DSR_E_IO, "Could not open file %s", fileName);
    const int sz = q->size_;
    auto wbe = [&](int v) { const unsigned u = (unsigned) v; const unsigned char b[4] = { (unsigned char) (u >> 24), (unsigned char) (u >> 16), (unsigned char) (u >> 8), (unsigned char) u }; fwrite(b, 1, 4, fp); };
    if (plainText) {
      fprintf(fp, "%d %d\n", q->frameX, sz);
      for (int i = 0; i <= q->frameX; i++) { const float* r = (const float*) q->row(i); for (int j = 0; j < sz; j++) { fprintf(fp, "%g", (double) r[j]); if (j < sz - 1) fprintf(fp, " "); } fprintf(fp, "\n"); }
    } else {
      wbe(q->frameX); wbe(sz);
      for (int i = 0; i <= q->frameX; i++) fwrite(q->row(i), sizeof(float), (size_t) sz, fp);
    }
    fclose(fp);
  });
}
dsr_status dsr_storage_read(dsr_stream* s, const char* fileName)
{
  return guard([&] {
    StorageOp* q = dynamic_cast<StorageOp*>(s); if (!q || !fileName) throw Error(DSR_E_PARAMETER, "not a StorageFeature");
    FILE* fp = fopen(fileName, "r"); if (!fp) throw Error(DSR_E_IO, "Could not open file %s", fileName);
    auto rbe = [&](int& v) { unsigned char b[4]; if (fread(b, 1, 4, fp) != 4) { fclose(fp); throw Error(DSR_E_IO, "premature end of %s", fileName); } v = (int) ((unsigned) b[0] << 24 | (unsigned) b[1] << 16 | (unsigned) b[2] << 8 | (unsigned) b[3]); };
    int fx = 0, sz = 0; rbe(fx); rbe(sz);
    if (sz != q->size_) { fclose(fp); throw Error(DSR_E_DIMENSION, "Feature dimensions (%d vs. %d) do not match.\n", sz, q->size_); }
    if (fx < 0 || fx >= 100000) { fclose(fp); throw Error(DSR_E_DIMENSION, "Frame %d is greater than maximum number %d.", fx, 100000); }
    // the operator now serves frames 0.._frameX from its own store: frames 0.._frameX-1 from the file, frame _frameX as the store had it (zero)
    q->nFrames = fx + 1; q->host.assign((size_t) q->nFrames * q->rowBytes() + 16, 0);
    for (int i = 0; i < fx; i++) if (fread(q->host.data() + (size_t) i * q->rowBytes(), sizeof(float), (size_t) sz, fp) != (size_t) sz) { fclose(fp); throw Error(DSR_E_IO, "premature end of %s", fileName); }
    fclose(fp);
    require_device(); q->dev.reserve(q->host.size()); DSR_HIP(hipMemcpy(q->dev.p, q->host.data(), (size_t) q->nFrames * q->rowBytes(), hipMemcpyHostToDevice));
    q->ready = true; q->frameX = fx; q->endOfSamples = false;
  });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ASR side of the boundary: the distribution set as the decoder sees it.  The reference decoder asks _dist->find(distX-1)->score(_frameX)
// (asr/decoder/decoder.h:985); Distrib::score -> CodebookBasic::score pulls frame frameX of the feature stream and caches the codebook's
// score for that frame (asr/gaussian/distribBasic.h:48-50,110-114, codebookBasic.cc:431-465).  Here a distribution set is a GMM model bound to
// a feature stream handle: score(distX, frameX) scores ALL distributions of that frame on the device at the first request and serves the rest
// of the frame's requests from the host copy; decode_stream() keeps everything on the device (features -> scores -> token passing), no host
// round trip.
struct dsr_distribset { dsr_gmm* gmm = nullptr; dsr_stream* feat = nullptr; int mode = 0; int cachedFrame = -1; std::vector<float> row; DevBuf<float> d_row, d_scores; DevBuf<int> d_T; };

dsr_status dsr_distribset_create(dsr_gmm* gmm, dsr_stream* feature, int gmmMode, dsr_distribset** out)
{
  return guard([&] {
    if (!gmm || !feature || !out) throw Error(DSR_E_PARAMETER, "null argument");
    if (feature->type != DSR_T_FLOAT) throw Error(DSR_E_TYPE, "the feature stream of a codebook set delivers float vectors");
    if (feature->size_ != dsr_gmm_dim(gmm)) throw Error(DSR_E_DIMENSION, "Feature and codebook dimensions (%d vs. %d) do not match.", feature->size_, dsr_gmm_dim(gmm));
    if (gmmMode < 0 || gmmMode > 2) throw Error(DSR_E_PARAMETER, "bad scoring mode %d", gmmMode);
    dsr_distribset* d = new dsr_distribset(); d->gmm = gmm; d->feat = feature; d->mode = gmmMode; dsr_stream_retain(feature); *out = d;
  });
}
void dsr_distribset_destroy(dsr_distribset* d) { if (d) { dsr_stream_release(d->feat); delete d; } }
int dsr_distribset_ndists(const dsr_distribset* d) { return d ? dsr_gmm_num_dists(d->gmm) : 0; }
dsr_status dsr_distribset_find(const dsr_distribset* d, const char* name, int* distX)
{ if (!d) return guard([&] { throw Error(DSR_E_PARAMETER, "null argument"); }); return dsr_gmm_find_dist(d->gmm, name, distX); }
const char* dsr_distribset_name(const dsr_distribset* d, int distX) { return d ? dsr_gmm_dist_name(d->gmm, distX) : ""; }
dsr_status dsr_distribset_reset_cache(dsr_distribset* d) { return guard([&] { if (!d) throw Error(DSR_E_PARAMETER, "null argument"); d->cachedFrame = -1; }); }      // codebookBasic.cc:414-420
dsr_status dsr_distribset_reset_feature(dsr_distribset* d) { return guard([&] { if (!d) throw Error(DSR_E_PARAMETER, "null argument"); d->feat->reset(); d->cachedFrame = -1; }); }
dsr_status dsr_distribset_score(dsr_distribset* d, int distX, int frameX, float* score)
{
  return guard([&] {
    if (!d || !score) throw Error(DSR_E_PARAMETER, "null argument");
    const int K = dsr_gmm_num_dists(d->gmm);
    if (distX < 0 || distX >= K) throw Error(DSR_E_INDEX, "distribution %d of %d", distX, K);
    if (frameX != d->cachedFrame || frameX < 0) {
      (void) d->feat->next(frameX);                                       // jiterator_error at the end of the stream, jindex_error out of order
      const int t = d->feat->frameX;
      d->d_row.reserve((size_t) K); d->row.resize((size_t) K);
      const float* x = reinterpret_cast<const float*>(d->feat->dev.p) + (size_t) t * d->feat->size_;
      const dsr_status s = dsr_gmm_score(d->gmm, x, 1, d->mode, d->d_row.p, nullptr, S0); if (s) throw Error(s, "%s", dsr_last_error());
      DSR_HIP(hipMemcpy(d->row.data(), d->d_row.p, sizeof(float) * (size_t) K, hipMemcpyDeviceToHost));
      d->cachedFrame = t;
    }
    *score = d->row[(size_t) distX];
  });
}
// _Decoder::decode() (decoder.h:688-737) for the utterance the feature stream currently holds: _newUtterance resets the cache and the feature
// (decoder.h:488-492), every frame is scored and decoded on the device.  An empty stream is DSR_E_ITERATOR (the exception escapes decode(), :691).
dsr_status dsr_decoder_decode_stream(dsr_decoder* dec, dsr_distribset* d, dsr_decode_result* res, int32_t* arcs_out, uint32_t* words_out, int maxPath)
{
  return guard([&] {
    if (!dec || !d || !res) throw Error(DSR_E_PARAMETER, "null argument");
    d->cachedFrame = -1; d->feat->reset();
    d->feat->materialize();
    const int T = d->feat->nFrames, K = dsr_gmm_num_dists(d->gmm);
    if (T <= 0) { d->feat->endOfSamples = true; throw Error(DSR_E_ITERATOR, "end of samples!"); }
    d->d_scores.reserve((size_t) T * K); d->d_T.upload(&T, 1);
    dsr_status s = dsr_gmm_score(d->gmm, reinterpret_cast<const float*>(d->feat->dev.p), (int64_t) T, d->mode, d->d_scores.p, nullptr, S0); if (s) throw Error(s, "%s", dsr_last_error());
    s = dsr_decoder_decode_batch(dec, d->d_scores.p, d->d_T.p, 1, T, K, res, arcs_out, words_out, maxPath, S0); if (s) throw Error(s, "%s", dsr_last_error());
    d->feat->frameX = T - 1; d->feat->endOfSamples = true;                   // the reference has pulled the stream to its end
    if (res->status != DSR_OK) throw Error(res->status, "decode failed (status %d)", res->status);
  });
}

// Lattice::gammaProbsDist(dss, acScale, lmScale, lmPenalty, silPenalty, silSymbol) (asr/lattice/lattice.cc:331-341): the links' acoustic scores are
// recomputed from the distribution set (_updateAc, :381-409: links of the initial node and of the nodes in _nodes -- not of the final nodes -- with
// an input symbol; score = sum over the link's frames), then gammaProbs.  The frames are scored once on the device, the per-link sums are a gather
// kernel over the score matrix.  A sum above LogZero is the reference's consistency error.
dsr_status dsr_lattice_gamma_probs_dist(dsr_lattice* L, dsr_distribset* d, double acScale, double lmScale, double lmPenalty, double silPenalty, unsigned silenceX,
                                        double* logProb)
{
  return guard([&] {
    if (!L || !d) throw Error(DSR_E_PARAMETER, "null argument");
    d->cachedFrame = -1;                                                     // dss->resetCache()
    L->ensure_ops(); L->sorted.clear();                                      // _clearSorted()
    d->feat->materialize();
    const int T = d->feat->nFrames, K = dsr_gmm_num_dists(d->gmm);
    std::vector<int> link, dist, start, end;
    auto take = [&](int node) {
      for (size_t k = 0; k < L->adj[(size_t) node].size(); k++) {
        const int e = L->adj[(size_t) node][k];
        if (L->in[(size_t) e] == 0) continue;
        const long dx = (long) L->in[(size_t) e] - 1;
        if (dx >= K) throw Error(DSR_E_INDEX, "link %d names distribution %ld of %d", e, dx, K);
        if (L->start[(size_t) e] <= L->end[(size_t) e] && (L->start[(size_t) e] < 0 || L->end[(size_t) e] >= T))
          throw Error(DSR_E_INDEX, "link %d spans frames %d..%d of %d", e, L->start[(size_t) e], L->end[(size_t) e], T);
        link.push_back(e); dist.push_back((int) dx); start.push_back(L->start[(size_t) e]); end.push_back(L->end[(size_t) e]);
      }
    };
    take(0);
    for (size_t p = 0; p < L->slots.size(); p++) if (L->slots[p] >= 0) take(L->slots[p]);
    if (!link.empty()) {
      if (T <= 0) throw Error(DSR_E_ITERATOR, "end of samples!");
      d->d_scores.reserve((size_t) T * K);
      const dsr_status s = dsr_gmm_score(d->gmm, reinterpret_cast<const float*>(d->feat->dev.p), (int64_t) T, d->mode, d->d_scores.p, nullptr, S0); if (s) throw Error(s, "%s", dsr_last_error());
      DevBuf<int> dd, ds, de; DevBuf<double> dout; dd.upload(dist); ds.upload(start); de.upload(end); dout.reserve(link.size());
      op_link_ac(d->d_scores.p, K, dd.p, ds.p, de.p, (int) link.size(), dout.p, S0);
      DSR_HIP(hipGetLastError());
      std::vector<double> sums(link.size());
      DSR_HIP(hipMemcpy(sums.data(), dout.p, sizeof(double) * link.size(), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < link.size(); i++) {
        if (sums[i] > 1.0E10) throw Error(DSR_E_CONSISTENCY, "Log-prob (%g) > LogZero (%g)", sums[i], 1.0E10);
        L->ac[(size_t) link[i]] = sums[i];
      }
    }
    const double p = L->gamma_probs(acScale, lmScale, lmPenalty, silPenalty, silenceX);
    if (logProb) *logProb = p;
  });
}

}  // extern "C"
