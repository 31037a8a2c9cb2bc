// csrc/k_postfilter.hip -- Zelinski post-filter on the beamformer output (first operator).
//
// Replaces calcCSD, TimeAlignment, ZelinskiFilter_f, ZelinskiFilter (btk/postfilter/postfilter.cc:8-221) and
// ZelinskiPostFilter::{setArrayManifoldVector,next} (:396-493) for halfBandShift == false.
//
// The auto/cross spectral densities of one bin are first-order recursions over the frames (forgetting factor alpha,
// alpha = 0 for the first two frames): sequential in time, independent across bins and utterances.  One thread owns one
// (utterance, bin) and walks the frames; its C(C+1)/2 densities live in a state array laid out [entry][utterance*F + bin],
// so that a wave's accesses to one entry are contiguous.  Arithmetic is fp64 in the reference's order (this file is
// compiled with -ffp-contract=off); the snapshots and the beamformer output are the pipe's complex64 arrays.
#include "common.h"
#include <cmath>

namespace dsr {

__global__ __launch_bounds__(128) void k_zelinski(const float2* __restrict__ X, const float2* __restrict__ Y, const int* __restrict__ nframesArr,
                                                  const double2* __restrict__ wq, double2* __restrict__ state, float2* __restrict__ out,
                                                  float* __restrict__ wp1, int U, int C, int Tmax, int F, double alphaCfg, int type, int minFrames)
{
  const long n = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (long) U * F) return;
  const int u = (int) (n / F), f = (int) (n - (long) u * F);
  const long S = (long) U * F;
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const float2* Xu = X + (long) u * C * Tmax * F;
  const float2* Yu = Y + (long) u * Tmax * F;
  float2* Ou = out + (long) u * Tmax * F;
  double2 ta[16];                                                // time-aligned channels (C <= 16 here; larger arrays: see the launcher)
  for (int t = 0; t < Tmax; t++) {
    if (t >= T) { Ou[(long) t * F + f] = make_float2(0.f, 0.f); if (wp1) wp1[((long) u * Tmax + t) * F + f] = 0.f; continue; }
    const int frameX = t - 1;                                    // _frameX before _increment() (postfilter.cc:463-476)
    const double alpha = (frameX > 0) ? alphaCfg : 0.0;
    const int pfType = (frameX < minFrames) ? 0 : type;
#pragma unroll 4
    for (int i = 0; i < C; i++) {                                // TimeAlignment: conj(d_i) x_i
      const double2 d = wq[(long) f * C + i]; const float2 x = Xu[((long) i * Tmax + t) * F + f];
      const double dr = d.x, di = -d.y, xr = (double) x.x, xi = (double) x.y;
      ta[i] = make_double2(dr * xr - di * xi, dr * xi + di * xr);
    }
    double sr = 0.0, si = 0.0; int e = 0;
    for (int i = 0; i < C - 1; i++)
      for (int j = i + 1; j < C; j++, e++) {
        const double ar = ta[i].x, ai = ta[i].y, br = ta[j].x, bi = -ta[j].y;
        const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
        double er = pr, ei = pi;
        if (alpha > 0.0) { const double2 p = state[(long) e * S + n]; er = p.x * alpha + pr * (1.0 - alpha); ei = p.y * alpha + pi * (1.0 - alpha); }
        sr += er; si += ei; state[(long) e * S + n] = make_double2(er, ei);
      }
    double numerator;
    if (1 & pfType) { numerator = sr; if (numerator < 0.0) numerator = 0.0; }
    else numerator = hypot(sr, si);
    double denominator = 0.0;
    for (int i = 0; i < C; i++, e++) {
      const double a2 = ta[i].x * ta[i].x + ta[i].y * ta[i].y;
      double est = a2;
      if (alpha > 0.0) est = alpha * state[(long) e * S + n].x + (1.0 - alpha) * a2;
      denominator += est; state[(long) e * S + n] = make_double2(est, 0.0);
    }
    double W = (numerator / denominator) * (2.0 / ((double) C - 1.0));
    if (W >= 1.0) W = 1.0;
    if (W < 0.0001) W = 0.0001;
    if (wp1) wp1[((long) u * Tmax + t) * F + f] = (float) W;
    const float2 y = Yu[(long) t * F + f];
    Ou[(long) t * F + f] = (pfType == 0) ? y : make_float2((float) (W * (double) y.x), (float) (W * (double) y.y));
  }
}

// the same with the densities in registers (C(C-1)/2 complex + C real fp64 values per thread), for the usual small arrays:
// no state traffic at all, the kernel then moves the algorithmic (C + 2) x 8 bytes per (frame, bin)
template <int C>
__global__ __launch_bounds__(64) void k_zelinski_reg(const float2* __restrict__ X, const float2* __restrict__ Y, const int* __restrict__ nframesArr,
                                                     const double2* __restrict__ wq, float2* __restrict__ out, float* __restrict__ wp1,
                                                     int U, int Tmax, int F, double alphaCfg, int type, int minFrames)
{
  constexpr int NP = C * (C - 1) / 2;
  const long n = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (long) U * F) return;
  const int u = (int) (n / F), f = (int) (n - (long) u * F);
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const float2* Xu = X + (long) u * C * Tmax * F;
  const float2* Yu = Y + (long) u * Tmax * F;
  float2* Ou = out + (long) u * Tmax * F;
  double2 d[C];
#pragma unroll
  for (int i = 0; i < C; i++) { d[i] = wq[(long) f * C + i]; d[i].y = -d[i].y; }
  double2 csd[NP]; double psd[C];
#pragma unroll
  for (int e = 0; e < NP; e++) csd[e] = make_double2(0.0, 0.0);
#pragma unroll
  for (int i = 0; i < C; i++) psd[i] = 0.0;
  for (int t = 0; t < Tmax; t++) {
    if (t >= T) { Ou[(long) t * F + f] = make_float2(0.f, 0.f); if (wp1) wp1[((long) u * Tmax + t) * F + f] = 0.f; continue; }
    const int frameX = t - 1;
    const double alpha = (frameX > 0) ? alphaCfg : 0.0;
    const int pfType = (frameX < minFrames) ? 0 : type;
    double2 ta[C];
#pragma unroll
    for (int i = 0; i < C; i++) {
      const float2 x = Xu[((long) i * Tmax + t) * F + f];
      const double xr = (double) x.x, xi = (double) x.y;
      ta[i] = make_double2(d[i].x * xr - d[i].y * xi, d[i].x * xi + d[i].y * xr);
    }
    double sr = 0.0, si = 0.0;
    {
      int e = 0;
#pragma unroll
      for (int i = 0; i < C - 1; i++)
#pragma unroll
        for (int j = i + 1; j < C; j++, e++) {
          const double ar = ta[i].x, ai = ta[i].y, br = ta[j].x, bi = -ta[j].y;
          const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
          double er = pr, ei = pi;
          if (alpha > 0.0) { er = csd[e].x * alpha + pr * (1.0 - alpha); ei = csd[e].y * alpha + pi * (1.0 - alpha); }
          sr += er; si += ei; csd[e] = make_double2(er, ei);
        }
    }
    double numerator;
    if (1 & pfType) { numerator = sr; if (numerator < 0.0) numerator = 0.0; }
    else numerator = hypot(sr, si);
    double denominator = 0.0;
#pragma unroll
    for (int i = 0; i < C; i++) {
      const double a2 = ta[i].x * ta[i].x + ta[i].y * ta[i].y;
      double est = a2;
      if (alpha > 0.0) est = alpha * psd[i] + (1.0 - alpha) * a2;
      denominator += est; psd[i] = est;
    }
    double W = (numerator / denominator) * (2.0 / ((double) C - 1.0));
    if (W >= 1.0) W = 1.0;
    if (W < 0.0001) W = 0.0001;
    if (wp1) wp1[((long) u * Tmax + t) * F + f] = (float) W;
    const float2 y = Yu[(long) t * F + f];
    Ou[(long) t * F + f] = (pfType == 0) ? y : make_float2((float) (W * (double) y.x), (float) (W * (double) y.y));
  }
}

struct ZelinskiPlan { int M = 0, C = 0, type = 2, minFrames = 0; double alpha = 0.6; std::vector<double> h_wq; bool dirty = true; DevBuf<double2> wq, state; };

}  // namespace dsr

using namespace dsr;
struct dsr_zelinski : ZelinskiPlan {};

extern "C" {

dsr_status dsr_zelinski_create(int fftLen, int chanN, double alpha, int type, int minFrames, dsr_zelinski** out)
{
  return guard([&] {
    if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (fftLen < 2 || (fftLen & 1)) throw Error(DSR_E_DIMENSION, "bad fftLen %d", fftLen);
    if (chanN <= 1) throw Error(DSR_E_DIMENSION, "The number of channels %d is <= 1", chanN);          // postfilter.cc:63-66
    if (chanN > 16) throw Error(DSR_E_DIMENSION, "post-filter kernel: at most 16 channels in this round (%d)", chanN);
    require_device();
    dsr_zelinski* p = new dsr_zelinski(); p->M = fftLen; p->C = chanN; p->alpha = alpha; p->type = type; p->minFrames = minFrames;
    p->h_wq.assign((size_t) (fftLen / 2 + 1) * chanN * 2, 0.0);
    *out = p;
  });
}
void dsr_zelinski_destroy(dsr_zelinski* p) { delete p; }
dsr_status dsr_zelinski_set_manifold(dsr_zelinski* p, int fbinX, const double* vec)
{
  return guard([&] {
    if (!p || !vec) throw Error(DSR_E_PARAMETER, "null argument");
    if (fbinX < 0 || fbinX >= p->M) throw Error(DSR_E_DIMENSION, "fbinX %d must be less than %d", fbinX, p->M);       // :398-401
    if (fbinX > p->M / 2) return;                                  // the mirror half is implied (halfBandShift == false)
    memcpy(&p->h_wq[(size_t) fbinX * p->C * 2], vec, sizeof(double) * 2 * p->C); p->dirty = true;
  });
}
dsr_status dsr_zelinski_apply(dsr_zelinski* p, const float* X, const float* Y, const int32_t* nframes_dev, int U, int Tmax, float* out, float* wp1, void* stream)
{
  return guard([&] {
    if (!p || !X || !Y || !nframes_dev || !out) throw Error(DSR_E_PARAMETER, "null argument");
    if (U <= 0 || Tmax <= 0) return;
    hipStream_t st = (hipStream_t) stream;
    const int F = p->M / 2 + 1;
    if (p->dirty) { std::vector<double2> w((size_t) F * p->C); for (size_t i = 0; i < w.size(); i++) w[i] = make_double2(p->h_wq[2 * i], p->h_wq[2 * i + 1]); p->wq.upload(w); p->dirty = false; }
    const size_t S = (size_t) U * F, NE = (size_t) p->C * (p->C + 1) / 2;
    const bool regs = !getenv("DSR_PF_MEMSTATE");
#define ZREG(CC) if (regs && p->C == CC) { hipLaunchKernelGGL(k_zelinski_reg<CC>, dim3((unsigned) ((S + 63) / 64)), dim3(64), 0, st, (const float2*) X, (const float2*) Y, \
      nframes_dev, p->wq.p, (float2*) out, wp1, U, Tmax, F, p->alpha, p->type, p->minFrames); DSR_HIP(hipGetLastError()); return; }
    ZREG(2) ZREG(3) ZREG(4) ZREG(6) ZREG(8)
#undef ZREG
    p->state.reserve(S * NE);
    hipLaunchKernelGGL(k_zelinski, dim3((unsigned) ((S + 127) / 128)), dim3(128), 0, st, (const float2*) X, (const float2*) Y, nframes_dev, p->wq.p, p->state.p,
                       (float2*) out, wp1, U, p->C, Tmax, F, p->alpha, p->type, p->minFrames);
    DSR_HIP(hipGetLastError());
  });
}

}  // extern "C"
