// csrc/capi.cpp -- error plumbing, device selection and the whole-pipe driver of the C-ABI (include/dsr.h).
#include "common.h"
#include <mutex>

namespace dsr {

static thread_local std::string g_lastError;
void set_last_error(const std::string& s) { g_lastError = s; }

void require_device()
{
  static thread_local int checked = 0;
  if (checked) return;
  int n = 0; hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw Error(DSR_E_INITIALIZATION, "no usable HIP device (%s): libdsr_hip has no CPU fallback", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  checked = 1;
}

}  // namespace dsr

using namespace dsr;

// the plan objects are defined in their kernels' translation units
struct dsr_fb; struct dsr_bf; struct dsr_mfcc; struct dsr_gmm; struct dsr_decoder;

struct dsr_pipe {
  const dsr_fb* ana; const dsr_fb* syn; dsr_bf* bf; dsr_mfcc* mfcc; dsr_gmm* gmm; dsr_decoder* dec; int gmmMode;
  DevBuf<float> X, Y, y, feat, scores; DevBuf<int> d_T, d_ny, d_Tm;
  std::vector<int> h_T, h_ny, h_Tm;
  hipEvent_t ev[7]; bool evInit = false; float ms[6] = {0, 0, 0, 0, 0, 0};
  int64_t bytes[5] = {0, 0, 0, 0, 0};
  bool pending = false, fused = false, ranFused = false;
};

extern "C" {

const char* dsr_last_error(void) { return g_lastError.c_str(); }
const char* dsr_version(void) { return "dsr-mi355x 0.1 (gfx950)"; }

dsr_status dsr_device_count(int* n)
{
  return guard([&] {
    if (!n) throw Error(DSR_E_PARAMETER, "null argument");
    int c = 0; hipError_t e = hipGetDeviceCount(&c); *n = (e == hipSuccess) ? c : 0;
  });
}
dsr_status dsr_set_device(int device) { return guard([&] { require_device(); DSR_HIP(hipSetDevice(device)); }); }
dsr_status dsr_stream_synchronize(void* stream) { return guard([&] { require_device(); DSR_HIP(hipStreamSynchronize((hipStream_t) stream)); }); }

dsr_status dsr_memcpy_dtoh(void* dst_host, const void* src_dev, size_t bytes, void* stream)
{
  return guard([&] {
    require_device();
    DSR_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t) stream));
    DSR_HIP(hipStreamSynchronize((hipStream_t) stream));
  });
}

dsr_status dsr_pipe_create(const dsr_fb* analysis, const dsr_fb* synthesis, dsr_bf* bf, dsr_mfcc* mfcc, dsr_gmm* gmm,
                           dsr_decoder* dec, int gmmMode, dsr_pipe** out)
{
  return guard([&] {
    if (!analysis || !synthesis || !bf || !mfcc || !gmm || !dec || !out) throw Error(DSR_E_PARAMETER, "null argument");
    if (dsr_mfcc_out_dim(mfcc) != dsr_gmm_dim(gmm)) throw Error(DSR_E_DIMENSION, "feature dimension %d != codebook dimension %d", dsr_mfcc_out_dim(mfcc), dsr_gmm_dim(gmm));
    require_device();
    dsr_pipe* p = new dsr_pipe(); p->ana = analysis; p->syn = synthesis; p->bf = bf; p->mfcc = mfcc; p->gmm = gmm; p->dec = dec; p->gmmMode = gmmMode;
    *out = p;
  });
}
void dsr_pipe_destroy(dsr_pipe* p) { if (p && p->evInit) for (int i = 0; i < 7; i++) (void) hipEventDestroy(p->ev[i]); delete p; }

static void check(dsr_status s) { if (s != DSR_OK) throw Error(s, "%s", dsr_last_error()); }

// submit enqueues the whole pipe for one batch on `stream` and returns; collect waits for it and hands the results out.
// A pipe object (and its decoder) carries one batch at a time; two pipe objects on two streams overlap the ragged end of
// one batch's decode with the front end of the next.
dsr_status dsr_pipe_submit(dsr_pipe* p, const float* x, const int32_t* nsamp_dev, const int32_t* nsamp_host, int U, int C,
                           int64_t sampStride, int maxPath, int want_paths, void* stream)
{
  return guard([&] {
    if (!p || !x || !nsamp_dev || !nsamp_host) throw Error(DSR_E_PARAMETER, "null argument");
    if (p->pending) throw Error(DSR_E_CONSISTENCY, "a batch is already in flight on this pipe: collect it first");
    if (U <= 0) return;
    hipStream_t st = (hipStream_t) stream;
    if (!p->evInit) { for (int i = 0; i < 7; i++) DSR_HIP(hipEventCreate(&p->ev[i])); p->evInit = true; }
    // host-side bookkeeping of the reference's frame counts (modulated.cc:461-516,626-664; feature.cc:610-659)
    p->h_T.resize(U); p->h_ny.resize(U); p->h_Tm.resize(U);
    int Tmax = 1, nyMax = 1, TmMax = 1;
    const int D = dsr_fb_block_len(p->syn);
    for (int u = 0; u < U; u++) {
      const int T = dsr_fb_analysis_frames(p->ana, nsamp_host[u]);
      p->h_T[u] = T; if (T > Tmax) Tmax = T;
      p->h_ny[u] = dsr_fb_synthesis_blocks(p->syn, T) * D;          // samples handed to SampleFeature::setSamples
      if (p->h_ny[u] > nyMax) nyMax = p->h_ny[u];
      p->h_Tm[u] = dsr_mfcc_frames(p->mfcc, p->h_ny[u]); if (p->h_Tm[u] > TmMax) TmMax = p->h_Tm[u];
    }
    if (dsr_bf_chan_n(p->bf) != C) throw Error(DSR_E_DIMENSION, "beamformer has %d channels, input has %d", dsr_bf_chan_n(p->bf), C);
    const int nDist = dsr_gmm_num_dists(p->gmm), dim = dsr_gmm_dim(p->gmm);
    const size_t F = (size_t) dsr_bf_fft_len(p->bf) / 2 + 1;
    if (!(p->fused && dsr_fb_analysis_beamform_supported(p->ana, p->bf))) p->X.reserve((size_t) U * C * Tmax * F * 2);
    p->Y.reserve((size_t) U * Tmax * F * 2); p->y.reserve((size_t) U * nyMax);
    p->feat.reserve((size_t) U * TmMax * dim); p->scores.reserve((size_t) U * TmMax * nDist);
    p->bytes[0] = (int64_t) U * C * Tmax * F * 8; p->bytes[1] = (int64_t) U * Tmax * F * 8; p->bytes[2] = (int64_t) U * nyMax * 4;
    p->bytes[3] = (int64_t) U * TmMax * dim * 4; p->bytes[4] = (int64_t) U * TmMax * nDist * 4;
    p->d_T.reserve(U); p->d_ny.reserve(U); p->d_Tm.reserve(U);
    DSR_HIP(hipMemcpyAsync(p->d_T.p, p->h_T.data(), sizeof(int) * U, hipMemcpyHostToDevice, st));
    DSR_HIP(hipMemcpyAsync(p->d_ny.p, p->h_ny.data(), sizeof(int) * U, hipMemcpyHostToDevice, st));
    DSR_HIP(hipMemcpyAsync(p->d_Tm.p, p->h_Tm.data(), sizeof(int) * U, hipMemcpyHostToDevice, st));

    DSR_HIP(hipEventRecord(p->ev[0], st));
    p->ranFused = p->fused && dsr_fb_analysis_beamform_supported(p->ana, p->bf);
    if (p->ranFused) {
      check(dsr_fb_analysis_beamform(p->ana, p->bf, x, nsamp_dev, U, C, sampStride, Tmax, p->Y.p, st));   // the snapshots X stay on the chip
      DSR_HIP(hipEventRecord(p->ev[1], st));
    } else {
      check(dsr_fb_analysis(p->ana, x, nsamp_dev, U, C, sampStride, Tmax, p->X.p, st));
      DSR_HIP(hipEventRecord(p->ev[1], st));
      check(dsr_bf_apply_frames(p->bf, p->X.p, p->d_T.p, U, Tmax, p->Y.p, st));     // an adapting beamformer stops at each utterance's last frame
    }
    DSR_HIP(hipEventRecord(p->ev[2], st));
    check(dsr_fb_synthesis(p->syn, p->Y.p, p->d_T.p, U, Tmax, nyMax, p->y.p, st));
    DSR_HIP(hipEventRecord(p->ev[3], st));
    check(dsr_mfcc_run(p->mfcc, p->y.p, p->d_ny.p, U, nyMax, TmMax, 0, p->feat.p, st));
    DSR_HIP(hipEventRecord(p->ev[4], st));
    check(dsr_gmm_score(p->gmm, p->feat.p, (int64_t) U * TmMax, p->gmmMode, p->scores.p, nullptr, st));
    DSR_HIP(hipEventRecord(p->ev[5], st));
    check(dsr_decoder_decode_launch(p->dec, p->scores.p, p->d_Tm.p, U, TmMax, nDist, maxPath, want_paths, st));
    DSR_HIP(hipEventRecord(p->ev[6], st));
    p->pending = true;
  });
}

dsr_status dsr_pipe_collect(dsr_pipe* p, dsr_decode_result* res, int32_t* arcs_out, uint32_t* words_out)
{
  return guard([&] {
    if (!p || !res) throw Error(DSR_E_PARAMETER, "null argument");
    if (!p->pending) throw Error(DSR_E_CONSISTENCY, "no batch in flight");
    p->pending = false;
    check(dsr_decoder_decode_collect(p->dec, res, arcs_out, words_out));
    DSR_HIP(hipEventSynchronize(p->ev[6]));
    for (int i = 0; i < 6; i++) DSR_HIP(hipEventElapsedTime(&p->ms[i], p->ev[i], p->ev[i + 1]));
  });
}

dsr_status dsr_pipe_run(dsr_pipe* p, const float* x, const int32_t* nsamp_dev, const int32_t* nsamp_host, int U, int C,
                        int64_t sampStride, dsr_decode_result* res, int32_t* arcs_out, uint32_t* words_out, int maxPath, void* stream)
{
  if (!res) return guard([&] { throw Error(DSR_E_PARAMETER, "null argument"); });
  if (U <= 0) return DSR_OK;
  const dsr_status s1 = dsr_pipe_submit(p, x, nsamp_dev, nsamp_host, U, C, sampStride, maxPath, (arcs_out || words_out) ? 1 : 0, stream);
  if (s1 != DSR_OK) return s1;
  return dsr_pipe_collect(p, res, arcs_out, words_out);
}

dsr_status dsr_pipe_set_fused(dsr_pipe* p, int fused)
{ return guard([&] { if (!p) throw Error(DSR_E_PARAMETER, "null argument"); if (p->pending) throw Error(DSR_E_CONSISTENCY, "a batch is in flight"); p->fused = fused != 0; }); }

dsr_status dsr_pipe_stage_ms(const dsr_pipe* p, float ms[6])
{ return guard([&] { if (!p || !ms) throw Error(DSR_E_PARAMETER, "null argument"); for (int i = 0; i < 6; i++) ms[i] = p->ms[i]; }); }

dsr_status dsr_pipe_intermediate(const dsr_pipe* p, int which, void** dev, int64_t* bytes)
{
  return guard([&] {
    if (!p || !dev) throw Error(DSR_E_PARAMETER, "null argument");
    if (which == 0 && p->ranFused) throw Error(DSR_E_CONSISTENCY, "the last run was fused: the channel snapshots were not written");
    void* q = which == 0 ? (void*) p->X.p : which == 1 ? (void*) p->Y.p : which == 2 ? (void*) p->y.p : which == 3 ? (void*) p->feat.p : which == 4 ? (void*) p->scores.p : nullptr;
    if (!q) throw Error(DSR_E_PARAMETER, "bad intermediate index %d", which);
    *dev = q; if (bytes) *bytes = p->bytes[which];
  });
}

}  // extern "C"
