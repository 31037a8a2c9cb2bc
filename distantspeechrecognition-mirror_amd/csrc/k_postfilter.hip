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
#include "svd_linpack.h"
#include <complex>
#include <cmath>

namespace dsr {

// ---------------------------------------------------------------------------------------------------------------------------------
// Zelinski for any array size, as two streaming kernels.  What the filter needs of the C (C - 1) / 2 cross densities is their SUM, and every
// density follows the same first-order recursion phi_ij(t) = alpha phi_ij(t - 1) + (1 - alpha) a_i(t) conj(a_j(t)) (postfilter.cc:8-21, 86-113):
// the sum obeys that recursion too, S(t) = alpha S(t - 1) + (1 - alpha) P(t), with P(t) = sum_{i<j} a_i conj(a_j) = sum_j (sum_{i<j} a_i) conj(a_j)
// -- a running prefix over the channels, O(C) complex products per (frame, bin) instead of O(C^2) densities kept and updated; the sum of the auto
// densities likewise with E(t) = sum_i |a_i|^2.  P and E of different frames are independent:
//   k_zel_pairs   one thread per (frame, bin), all frames in parallel: reads every snapshot once, fully coalesced ([frame][bin] is contiguous per
//                 channel), the time-alignment vector from its transposed copy [channel][bin] (the waves of a CU walk the channels together: L1);
//   k_zel_recur   the two scalar recursions, the weight, the filtered output: sixteen threads per (stream, bin), a stretch of the frames each (see the kernel).
// fp64 throughout, as the reference; the sums are the reference's numbers up to the rounding of a different summation order (1e-13 relative:
// tests compare at 1e-6).  The carried state of a stream (block streaming) is S and the auto sum: three doubles per bin instead of C (C + 1) / 2
// complex densities.  At 64 channels x 32 streams x 1250 frames: 24.0 -> 1.6 ms (the wave-per-bin kernel that kept all 2016 densities was bound
// by its fp64 pair updates, 1.4 % of HBM); the kernel is now bound by reading the snapshots.
// Carried state (block streaming, dsr_zelinski_carry): seenIn[u] = frames of stream u the earlier calls have filtered (the recursions start from
// scratch only on the stream's own first two frames), seenOut[u] = that plus this call's; the sums are read from / left in `state`.
// BF: the same pass also forms the beamformer's output Y = sum_c conj(w_c) x_c (SubbandDS / MVDR::next, beamformer.cc:1159-1175; the operation and its order are
// k_bf_apply's) from the channel-major weight image of the beamformer the post-filter sits behind (ZelinskiPostFilter::setBeamformer, postfilter.cc:376) -- beamformer
// and post-filter read the snapshots once instead of twice (64 channels x 32 streams x 1250 frames: 2.7 GB, 0.65 ms at the rate either pass reaches).
// (A thread takes TWO frames of a bin, t and t + ceil(Tmax / 2): the time-alignment vector and the beamformer's weight of a channel are fetched once for both.)
template <bool BF>
__global__ __launch_bounds__(256) void k_zel_pairs(const float2* __restrict__ X, const int* __restrict__ nframesArr, const double2* __restrict__ wqT,
                                                   double2* __restrict__ Pb, double* __restrict__ Eb, int C, int Tmax, int F,
                                                   const float2* __restrict__ wT, int wPitch, float2* __restrict__ Yout)
{
  const long TF = (long) Tmax * F;
  const int H = (Tmax + 1) / 2;                                   // frames of the first half
  const long idx = (long) blockIdx.x * 256 + threadIdx.x;         // (frame of the first half, bin)
  const int u = blockIdx.y;
  if (idx >= (long) H * F) return;
  const int t = (int) (idx / F), f = (int) (idx - (long) t * F);
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const int t2 = t + H; const long idx2 = idx + (long) H * F;
  const bool on1 = t < T, on2 = t2 < T, in2 = t2 < Tmax;
  if (BF) { if (!on1) Yout[(long) u * TF + idx] = make_float2(0.f, 0.f); if (in2 && !on2) Yout[(long) u * TF + idx2] = make_float2(0.f, 0.f); }   // (the padded rows of the snapshots are zero, so is the beamformer's output there)
  if (!on1) return;                                               // (t2 > t: nothing to do for the second frame either)
  const float2* Xp = X + (long) u * C * TF + idx;
  const long o2 = on2 ? (long) H * F : 0;                         // (an idle second frame re-reads the first: its sums are dropped)
  const double2* dp = wqT + f;
  double Ar = 0.0, Ai = 0.0, Pr = 0.0, Pi = 0.0, E = 0.0; float2 acc = make_float2(0.f, 0.f);
  double Br = 0.0, Bi = 0.0, Qr = 0.0, Qi = 0.0, G = 0.0; float2 acd = make_float2(0.f, 0.f);
#pragma unroll 4
  for (int c = 0; c < C; c++) {
    const float2 x = Xp[(long) c * TF], z = Xp[(long) c * TF + o2]; const double2 d = dp[(long) c * F];
    if (BF) {
      const float2 wc = wT[(long) c * wPitch + f];
      acc.x += wc.x * x.x + wc.y * x.y; acc.y += wc.x * x.y - wc.y * x.x;           // conj(w) x
      acd.x += wc.x * z.x + wc.y * z.y; acd.y += wc.x * z.y - wc.y * z.x;
    }
    const double dr = d.x, di = -d.y;
    { const double xr = (double) x.x, xi = (double) x.y;
      const double ar = dr * xr - di * xi, ai = dr * xi + di * xr;        // TimeAlignment: conj(d_c) x_c (postfilter.cc:30-43)
      Pr += Ar * ar + Ai * ai; Pi += Ai * ar - Ar * ai;                   // (sum of the channels before) conj(a_c)
      E += ar * ar + ai * ai; Ar += ar; Ai += ai; }
    { const double xr = (double) z.x, xi = (double) z.y;
      const double ar = dr * xr - di * xi, ai = dr * xi + di * xr;
      Qr += Br * ar + Bi * ai; Qi += Bi * ar - Br * ai;
      G += ar * ar + ai * ai; Br += ar; Bi += ai; }
  }
  Pb[(long) u * TF + idx] = make_double2(Pr, Pi); Eb[(long) u * TF + idx] = E;
  if (BF) Yout[(long) u * TF + idx] = acc;
  if (on2) { Pb[(long) u * TF + idx2] = make_double2(Qr, Qi); Eb[(long) u * TF + idx2] = G; if (BF) Yout[(long) u * TF + idx2] = acd; }
}

// One workgroup = 64 (stream, bin) series x 16 stretches of the time axis (a wave per stretch; lanes = neighbouring bins: every load is one contiguous run).
// A step of either recursion is an affine map s -> a s + b (a = alpha, or 0 where the reference restarts it), maps compose, so a stretch is first reduced to its
// own map from the sums alone (no weight, no output), the sixteen maps are chained through LDS, and every stretch then runs again from the state it really starts
// in -- hypot, division, clamps and the output, the expensive part, sixteen stretches at once.  (One thread a series, 4128 threads walking 1250 frames: 0.9 ms, then
// 0.5 ms with sixteen frames' loads in flight; this way the series of a stream are 66 k threads.)  The states differ from the frame-by-frame walk by the rounding
// of a different association (1e-16 relative).
__global__ __launch_bounds__(1024) void k_zel_recur(const double2* __restrict__ Pb, const double* __restrict__ Eb, const float2* __restrict__ Y,
                                                    const int* __restrict__ nframesArr, double2* __restrict__ state, float2* __restrict__ out,
                                                    float* __restrict__ wp1, int U, int C, int Tmax, int F, double alphaCfg, int type, int minFrames,
                                                    const int* __restrict__ seenIn, int* __restrict__ seenOut, int carryIn, int carryOut)
{
  constexpr int NST = 16;                                        // stretches = waves of the workgroup
  __shared__ double s_a[NST][64], s_br[NST][64], s_bi[NST][64], s_bd[NST][64];
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  const long n = (long) blockIdx.x * 64 + lane;
  const long S = (long) U * F;
  const bool live = n < S;
  const int u = live ? (int) (n / F) : 0, f = live ? (int) (n - (long) u * F) : 0;
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const int seen = seenIn ? seenIn[u] : 0;
  if (seenOut && live && f == 0 && k == 0) seenOut[u] = seen + T;
  const int L = (Tmax + NST - 1) / NST, t0 = k * L, t1 = (t0 + L < Tmax) ? t0 + L : Tmax;
  const long base = (long) u * Tmax * F + f;
  // ---- the stretch as a map: (a, b) with b = (Sr, Si, D) reached from zero
  double a = 1.0, br = 0.0, bi = 0.0, bd = 0.0;
  if (live) {
    constexpr int PF = 8;
    for (int tb = t0; tb < t1 && tb < T; tb += PF) {
      double2 Pv[PF]; double Ev[PF];
#pragma unroll
      for (int i = 0; i < PF; i++) { const int t = tb + i < T ? tb + i : T - 1; const long o = base + (long) t * F; Pv[i] = Pb[o]; Ev[i] = Eb[o]; }
#pragma unroll
      for (int i = 0; i < PF; i++) {
        const int t = tb + i; if (t >= t1 || t >= T) break;
        const double alpha = (seen + t - 1 > 0) ? alphaCfg : 0.0;
        if (alpha > 0.0) { br = br * alpha + Pv[i].x * (1.0 - alpha); bi = bi * alpha + Pv[i].y * (1.0 - alpha); bd = alpha * bd + (1.0 - alpha) * Ev[i]; a *= alpha; }
        else { br = Pv[i].x; bi = Pv[i].y; bd = Ev[i]; a = 0.0; }
      }
    }
  }
  s_a[k][lane] = a; s_br[k][lane] = br; s_bi[k][lane] = bi; s_bd[k][lane] = bd;
  __syncthreads();
  // ---- the state this stretch starts in: the stream's carried state through the maps of the stretches before
  double Sr = 0.0, Si = 0.0, D = 0.0;
  if (live && carryIn) { const double2 s0 = state[n]; Sr = s0.x; Si = s0.y; D = state[S + n].x; }
  for (int j = 0; j < k; j++) { const double aj = s_a[j][lane]; Sr = aj * Sr + s_br[j][lane]; Si = aj * Si + s_bi[j][lane]; D = aj * D + s_bd[j][lane]; }
  if (!live) return;
  const double scale = 2.0 / ((double) C - 1.0);
  constexpr int PF = 8;
  for (int tb = t0; tb < t1; tb += PF) {
    double2 Pv[PF]; double Ev[PF]; float2 yv[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) { const int t = tb + i < T ? tb + i : (T > 0 ? T - 1 : 0); const long o = base + (long) t * F; Pv[i] = Pb[o]; Ev[i] = Eb[o]; yv[i] = Y[o]; }
#pragma unroll
    for (int i = 0; i < PF; i++) {
      const int t = tb + i; if (t >= t1) break;
      const long o = base + (long) t * F;
      if (t >= T) { out[o] = make_float2(0.f, 0.f); if (wp1) wp1[o] = 0.f; continue; }
      const double2 P = Pv[i]; const double E = Ev[i]; const float2 y = yv[i];
      const int frameX = seen + t - 1;                                    // _frameX before _increment() (postfilter.cc:463-466)
      const double alpha = (frameX > 0) ? alphaCfg : 0.0;
      if (alpha > 0.0) { Sr = Sr * alpha + P.x * (1.0 - alpha); Si = Si * alpha + P.y * (1.0 - alpha); D = alpha * D + (1.0 - alpha) * E; }
      else { Sr = P.x; Si = P.y; D = E; }
      const int pfType = (frameX < minFrames) ? 0 : type;
      double numerator;
      if (1 & pfType) { numerator = Sr; if (numerator < 0.0) numerator = 0.0; } else numerator = hypot(Sr, Si);
      double W = (numerator / D) * scale;
      if (W >= 1.0) W = 1.0;
      if (W < 0.0001) W = 0.0001;
      if (wp1) wp1[o] = (float) W;
      out[o] = (pfType == 0) ? y : make_float2((float) (W * (double) y.x), (float) (W * (double) y.y));
    }
  }
  if (carryOut && k == NST - 1) { state[n] = make_double2(Sr, Si); state[S + n] = make_double2(D, 0.0); }
}

// McCowanPostFilter (postfilter.cc:706-744,789-826,833-945): the same recursions, then the noise-coherence corrected estimate of the
// clean-signal PSD averaged over the microphone pairs.  R: [F][C][C] noise coherence (fp64 complex), thr = _thresholdOfRij.
__device__ __forceinline__ double2 cdiv_gsl(double ar, double ai, double br, double bi)
{ const double s = 1.0 / hypot(br, bi); const double sbr = s * br, sbi = s * bi; return make_double2((ar * sbr + ai * sbi) * s, (ai * sbr - ar * sbi) * s); }

// (CT: compile-time channel count -- the densities then live in registers, C(C+1)/2 complex fp64 values per thread, and the kernel moves only
// the algorithmic bytes; 0: run-time count, densities in the state array [entry][utterance x bin])
template <int CT>
__global__ __launch_bounds__(128) void k_mccowan(const float2* __restrict__ X, const float2* __restrict__ Y, const int* __restrict__ nframesArr,
                                                 const double2* __restrict__ wq, const double2* __restrict__ R, double2* __restrict__ state,
                                                 float2* __restrict__ out, float* __restrict__ wp1, int U, int Crt, int Tmax, int F, double alphaCfg, int type,
                                                 int minFrames, double thr, const double2* __restrict__ lambda, int fbinX1,
                                                 const int* __restrict__ seenIn, int* __restrict__ seenOut, int carryIn, int carryOut)
{
  // lambda != nullptr: LefkimmiatisPostFilter (postfilter.cc:1065-1176) -- McCowan's clean-signal estimate against the noise estimate
  // sum (0.5 (phi_ii + phi_jj) - phi_ij) / (1 - R_ij), divided by d^H pinv(R) d from bin fbinX1 on
  const int C = CT ? CT : Crt;
  const long n = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (long) U * F) return;
  const int u = (int) (n / F), f = (int) (n - (long) u * F);
  const long S = (long) U * F;
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const float2* Xu = X + (long) u * C * Tmax * F;
  const float2* Yu = Y + (long) u * Tmax * F;
  float2* Ou = out + (long) u * Tmax * F;
  const double2* Rf = R + (long) f * C * C;
  const int NPp = C * (C - 1) / 2;
  constexpr int CA = CT ? CT : 16;
  double2 ta[CA]; double psd[CA];
  double2 stR[CT ? CT * (CT + 1) / 2 : 1];
#define ST(e) (CT ? stR[CT ? (e) : 0] : state[(long) (e) * S + n])
  const int seen = seenIn ? seenIn[u] : 0;
  if (seenOut && f == 0) seenOut[u] = seen + T;
  if (CT && carryIn) for (int e = 0; e < (CT ? CT * (CT + 1) / 2 : 0); e++) stR[CT ? e : 0] = state[(long) e * S + n];
  for (int t = 0; t < Tmax; t++) {
    if (t >= T) { Ou[(long) t * F + f] = make_float2(0.f, 0.f); if (wp1) wp1[((long) u * Tmax + t) * F + f] = 0.f; continue; }
    const int frameX = seen + t - 1;
    const double alpha = (frameX > 0) ? alphaCfg : 0.0;
    for (int i = 0; i < C; i++) {
      const double2 d = wq[(long) f * C + i]; const float2 x = Xu[((long) i * Tmax + t) * F + f];
      const double dr = d.x, di = -d.y, xr = (double) x.x, xi = (double) x.y;
      ta[i] = make_double2(dr * xr - di * xi, dr * xi + di * xr);
    }
    int e = 0;
    for (int i = 0; i < C - 1; i++)
      for (int j = i + 1; j < C; j++, e++) {
        const double ar = ta[i].x, ai = ta[i].y, br = ta[j].x, bi = -ta[j].y;
        const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
        double er = pr, ei = pi;
        if (alpha > 0.0) { const double2 p = ST(e); er = p.x * alpha + pr * (1.0 - alpha); ei = p.y * alpha + pi * (1.0 - alpha); }
        ST(e) = make_double2(er, ei);
      }
    double sumOfPSD = 0.0;
    for (int i = 0; i < C; i++) {
      const double a2 = ta[i].x * ta[i].x + ta[i].y * ta[i].y;
      double est = a2;
      if (alpha > 0.0) est = alpha * ST(NPp + i).x + (1.0 - alpha) * a2;
      sumOfPSD += est; psd[i] = est; ST(NPp + i) = make_double2(est, 0.0);
    }
    const double de = sumOfPSD / (double) C;
    double sr = 0.0, si = 0.0; e = 0;
    for (int i = 0; i < C - 1; i++)
      for (int j = i + 1; j < C; j++, e++) {
        const double2 phi = ST(e);
        double2 r = Rf[i * C + j];
        if (r.x > thr && r.y <= 0.0) r = make_double2(thr, 0.0);
        const double hs = 0.5 * (psd[i] + psd[j]);
        const double2 q = cdiv_gsl(phi.x - r.x * hs, phi.y - r.y * hs, -r.x + 1.0, -r.y);
        sr += q.x; si += q.y;
      }
    const double avg = (1 & type) ? sr : hypot(sr, si);
    const double nu = 2.0 * avg / (double) (C * (C - 1));
    double W = nu / de;
    if (lambda) {
      double vr = 0.0, vi = 0.0; e = 0;
      for (int i = 0; i < C - 1; i++)
        for (int j = i + 1; j < C; j++, e++) {
          const double2 phi = ST(e);
          double2 r = Rf[i * C + j];
          if (r.x > thr) r = make_double2(thr, 0.0); else if (r.x == 1.0) r = make_double2(0.99, 0.0);
          const double2 q = cdiv_gsl((psd[i] + psd[j]) * 0.5 - phi.x, 0.0 - phi.y, -r.x + 1.0, -r.y);
          vr += q.x; vi += q.y;
        }
      const double phi_vv = 2.0 * ((1 & type) ? vr : hypot(vr, vi)) / (double) (C * (C - 1));
      if (f < fbinX1) W = nu / (nu + phi_vv);
      else { const double2 l = lambda[f]; W = nu / (nu + phi_vv / ((1 & type) ? l.x : hypot(l.x, l.y))); }
    }
    if (W > 1.0) W = 1.0;
    if (W < 0.0001) W = 0.0001;
    if (wp1) wp1[((long) u * Tmax + t) * F + f] = (float) W;
    const float2 y = Yu[(long) t * F + f];
    Ou[(long) t * F + f] = (frameX >= minFrames) ? make_float2((float) ((double) y.x * W), (float) ((double) y.y * W)) : y;
  }
  if (CT && carryOut) for (int e = 0; e < (CT ? CT * (CT + 1) / 2 : 0); e++) state[(long) e * S + n] = stR[CT ? e : 0];
}
#undef ST

// the same with the densities in registers (C(C-1)/2 complex + C real fp64 values per thread), for the usual small arrays:
// no state traffic at all, the kernel then moves the algorithmic (C + 2) x 8 bytes per (frame, bin)
template <int C>
__global__ __launch_bounds__(64) void k_zelinski_reg(const float2* __restrict__ X, const float2* __restrict__ Y, const int* __restrict__ nframesArr,
                                                     const double2* __restrict__ wq, float2* __restrict__ out, float* __restrict__ wp1,
                                                     int U, int Tmax, int F, double alphaCfg, int type, int minFrames,
                                                     const int* __restrict__ seenIn, int* __restrict__ seenOut, double2* __restrict__ state, int carryIn, int carryOut)
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
  const long S = (long) U * F;
  const int seen = seenIn ? seenIn[u] : 0;
  if (seenOut && f == 0) seenOut[u] = seen + T;
  if (carryIn) {
#pragma unroll
    for (int e = 0; e < NP; e++) csd[e] = state[(long) e * S + n];
#pragma unroll
    for (int i = 0; i < C; i++) psd[i] = state[(long) (NP + i) * S + n].x;
  }
  for (int t = 0; t < Tmax; t++) {
    if (t >= T) { Ou[(long) t * F + f] = make_float2(0.f, 0.f); if (wp1) wp1[((long) u * Tmax + t) * F + f] = 0.f; continue; }
    const int frameX = seen + t - 1;
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
  if (carryOut) {
#pragma unroll
    for (int e = 0; e < NP; e++) state[(long) e * S + n] = csd[e];
#pragma unroll
    for (int i = 0; i < C; i++) state[(long) (NP + i) * S + n] = make_double2(psd[i], 0.0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Large arrays (16 < C <= 64, BASELINE configs[4]: 64 channels = 2016 microphone pairs): one WAVEFRONT per (utterance, bin).  Lane c time-aligns
// channel c and owns its auto-density; the cross densities are dealt round-robin over the lanes (pair e belongs to lane e mod 64: at most 32 per
// lane, in registers together with the pair's noise coherence); the time-aligned snapshot of the frame goes through a 1.5 KB LDS strip of the
// wave; the sums over pairs and channels are wave reductions (fp64 adds in a tree, not in the reference's pair order: 1e-15 relative).
// One kernel for the three filters: kind 0 Zelinski, 1 McCowan, 2 Lefkimmiatis (lambda != nullptr).
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int KIND>
__global__ __launch_bounds__(64) void k_pf_wave(const float2* __restrict__ X, const float2* __restrict__ Y, const int* __restrict__ nframesArr,
                                                const double2* __restrict__ wq, const double2* __restrict__ R, const unsigned short* __restrict__ pairIJ,
                                                double2* __restrict__ state, float2* __restrict__ out, float* __restrict__ wp1, int U, int C, int Tmax, int F,
                                                double alphaCfg, int type, int minFrames, double thr, const double2* __restrict__ lambda, int fbinX1,
                                                const int* __restrict__ seenIn, int* __restrict__ seenOut, int carryIn, int carryOut)
{
  constexpr int KP = 32;                                         // pairs per lane (64 channels: 2016 = 31.5 x 64)
  __shared__ double2 taS[64]; __shared__ double psdS[64];
  const long n = blockIdx.x; const int lane = threadIdx.x;
  const int u = (int) (n / F), f = (int) (n - (long) u * F);
  const long S = (long) U * F;
  const int NP = C * (C - 1) / 2;
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const float2* Xl = X + ((long) u * C + (lane < C ? lane : 0)) * Tmax * F + f;
  const float2* Yu = Y + (long) u * Tmax * F;
  float2* Ou = out + (long) u * Tmax * F;
  const int seen = seenIn ? seenIn[u] : 0;
  if (seenOut && f == 0 && lane == 0) seenOut[u] = seen + T;
  double2 dconj = make_double2(0.0, 0.0);
  if (lane < C) { dconj = wq[(long) f * C + lane]; dconj.y = -dconj.y; }
  // the pairs' noise coherences (McCowan / Lefkimmiatis) wait in LDS, [pair]: 32 KB at 64 channels, read as lane-contiguous 16-byte words
  extern __shared__ __attribute__((aligned(16))) unsigned char smemPf[];
  double2* rrS = reinterpret_cast<double2*>(smemPf);
  unsigned short ij[KP]; double2 csd[KP];
#pragma unroll
  for (int k = 0; k < KP; k++) {
    const int e = lane + 64 * k;
    ij[k] = e < NP ? pairIJ[e] : (unsigned short) 0;
    csd[k] = (carryIn && e < NP) ? state[(long) e * S + n] : make_double2(0.0, 0.0);
    if (KIND && e < NP) rrS[e] = R[(long) f * C * C + (ij[k] & 255) * C + (ij[k] >> 8)];
  }
  double psdSt = (carryIn && lane < C) ? state[(long) (NP + lane) * S + n].x : 0.0;
  float2 xn = (T > 0 && lane < C) ? Xl[0] : make_float2(0.f, 0.f);
  for (int t = 0; t < Tmax; t++) {
    if (t >= T) { if (lane == 0) { Ou[(long) t * F + f] = make_float2(0.f, 0.f); if (wp1) wp1[((long) u * Tmax + t) * F + f] = 0.f; } continue; }
    const float2 x = xn;
    if (t + 1 < T && lane < C) xn = Xl[(long) (t + 1) * F];       // the next frame's snapshot is in flight while this one is worked on
    const int frameX = seen + t - 1;
    const double alpha = (frameX > 0) ? alphaCfg : 0.0;
    const double xr = (double) x.x, xi = (double) x.y;
    const double2 ta = make_double2(dconj.x * xr - dconj.y * xi, dconj.x * xi + dconj.y * xr);
    double est = 0.0;
    if (lane < C) {
      const double a2 = ta.x * ta.x + ta.y * ta.y;
      est = (alpha > 0.0) ? alpha * psdSt + (1.0 - alpha) * a2 : a2;
      psdSt = est; taS[lane] = ta; psdS[lane] = est;
    }
    __syncthreads();                                             // one wave per workgroup: an LDS fence
    double sr = 0.0, si = 0.0, vr = 0.0, vi = 0.0;
#pragma unroll
    for (int k = 0; k < KP; k++) {
      const int e = lane + 64 * k;
      if (e < NP) {
        const int i = ij[k] & 255, j = ij[k] >> 8;
        const double2 a = taS[i], b = taS[j];
        const double pr = a.x * b.x + a.y * b.y, pi = a.y * b.x - a.x * b.y;          // a conj(b)
        double er = pr, ei = pi;
        if (alpha > 0.0) { er = csd[k].x * alpha + pr * (1.0 - alpha); ei = csd[k].y * alpha + pi * (1.0 - alpha); }
        csd[k] = make_double2(er, ei);
        if (KIND == 0) { sr += er; si += ei; }
        else {
          const double2 r0 = rrS[e];
          double2 r = r0;
          if (r.x > thr && r.y <= 0.0) r = make_double2(thr, 0.0);
          const double hs = 0.5 * (psdS[i] + psdS[j]);
          const double2 q = cdiv_gsl(er - r.x * hs, ei - r.y * hs, -r.x + 1.0, -r.y);
          sr += q.x; si += q.y;
          if (KIND == 2) {
            double2 r2 = r0;
            if (r2.x > thr) r2 = make_double2(thr, 0.0); else if (r2.x == 1.0) r2 = make_double2(0.99, 0.0);
            const double2 q2 = cdiv_gsl(hs - er, 0.0 - ei, -r2.x + 1.0, -r2.y);
            vr += q2.x; vi += q2.y;
          }
        }
      }
    }
    __syncthreads();
    sr = wave_sum(sr); si = wave_sum(si);
    const double sumPSD = wave_sum(est);
    if (KIND == 2) { vr = wave_sum(vr); vi = wave_sum(vi); }
    if (lane == 0) {
      double W; bool pass;
      if (KIND == 0) {
        const int pfType = (frameX < minFrames) ? 0 : type;
        double numerator;
        if (1 & pfType) { numerator = sr; if (numerator < 0.0) numerator = 0.0; } else numerator = hypot(sr, si);
        W = (numerator / sumPSD) * (2.0 / ((double) C - 1.0));
        if (W >= 1.0) W = 1.0;
        pass = pfType == 0;
      } else {
        const double de = sumPSD / (double) C;
        const double avg = (1 & type) ? sr : hypot(sr, si);
        const double nu = 2.0 * avg / (double) (C * (C - 1));
        W = nu / de;
        if (KIND == 2) {
          const double phi_vv = 2.0 * ((1 & type) ? vr : hypot(vr, vi)) / (double) (C * (C - 1));
          if (f < fbinX1) W = nu / (nu + phi_vv);
          else { const double2 l = lambda[f]; W = nu / (nu + phi_vv / ((1 & type) ? l.x : hypot(l.x, l.y))); }
        }
        if (W > 1.0) W = 1.0;
        pass = !(frameX >= minFrames);
      }
      if (W < 0.0001) W = 0.0001;
      if (wp1) wp1[((long) u * Tmax + t) * F + f] = (float) W;
      const float2 y = Yu[(long) t * F + f];
      Ou[(long) t * F + f] = pass ? y : make_float2((float) (W * (double) y.x), (float) (W * (double) y.y));
    }
  }
  if (carryOut) {
#pragma unroll
    for (int k = 0; k < KP; k++) { const int e = lane + 64 * k; if (e < NP) state[(long) e * S + n] = csd[k]; }
    if (lane < C) state[(long) (NP + lane) * S + n] = make_double2(psdSt, 0.0);
  }
}

struct ZelinskiPlan { int M = 0, C = 0, type = 2, minFrames = 0; double alpha = 0.6; std::vector<double> h_wq; bool dirty = true; DevBuf<double2> wq, state;
                      int kind = 0; double threshold = 0.99; std::vector<double> h_R; bool haveR = false, dirtyR = true; DevBuf<double2> R;       // kind 1: McCowan
                      double minSV = 1e-8; int fbinX1 = 0; bool dirtyL = true; DevBuf<double2> lambda;                                                 // kind 2: Lefkimmiatis
                      DevBuf<double2> wqT; struct PE { DevBuf<double2> P; DevBuf<double> E; DevBuf<float2> Y; }; PerStream<PE> pe;                                          // Zelinski: [chan][bin] manifold, P / E of a call (one set per stream)
                      DevBuf<unsigned short> pairIJ;                                                                                                      // wave kernel: pair e -> i | j << 8
                      bool carry = false, haveState = false; int stateU = 0; DevBuf<int> seen[2]; int seenCur = 0; };                                      // carried state (block streaming)

}  // namespace dsr

using namespace dsr;
struct dsr_zelinski : ZelinskiPlan {};


// calcInverseNoiseSpatialSpectralMatrix + calcLambda (postfilter.cc:980-1009): invR = pseudoinverse(R, minSV) -- LINPACK csvdc in
// complex<float>, restated in svd_linpack.cpp -- replaced by the identity when a singular value falls below minSV; tmpH = invR^H d,
// Lambda = tmpH^H d in double.
static double2 lefkimmiatis_lambda(const double* Rf, const double* df, int C, double minSV)
{
  typedef std::complex<double> cd;
  std::vector<cd> inv((size_t) C * C), tmpH(C);
  if (!linpack::pseudoinverse(reinterpret_cast<const cd*>(Rf), inv.data(), C, C, (float) minSV)) {
    std::fill(inv.begin(), inv.end(), cd(0.0, 0.0));
    for (int i = 0; i < C; i++) inv[(size_t) i * C + i] = cd(1.0, 0.0);
  }
  for (int i = 0; i < C; i++) { cd a(0.0, 0.0); for (int j = 0; j < C; j++) a += std::conj(inv[(size_t) j * C + i]) * cd(df[2 * j], df[2 * j + 1]); tmpH[i] = a; }
  cd lam(0.0, 0.0);
  for (int i = 0; i < C; i++) lam += std::conj(tmpH[i]) * cd(df[2 * i], df[2 * i + 1]);
  return make_double2(lam.real(), lam.imag());
}

extern "C" {

dsr_status dsr_zelinski_create(int fftLen, int chanN, double alpha, int type, int minFrames, dsr_zelinski** out)
{
  return guard([&] {
    if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (fftLen < 2 || (fftLen & 1)) throw Error(DSR_E_DIMENSION, "bad fftLen %d", fftLen);
    if (chanN <= 1) throw Error(DSR_E_DIMENSION, "The number of channels %d is <= 1", chanN);          // postfilter.cc:63-66
    if (chanN > 64) throw Error(DSR_E_DIMENSION, "post-filter kernels: at most 64 channels (%d)", chanN);
    require_device();
    dsr_zelinski* p = new dsr_zelinski(); p->M = fftLen; p->C = chanN; p->alpha = alpha; p->type = type; p->minFrames = minFrames;
    p->h_wq.assign((size_t) (fftLen / 2 + 1) * chanN * 2, 0.0);
    *out = p;
  });
}
void dsr_zelinski_destroy(dsr_zelinski* p) { delete p; }

// McCowanPostFilter (postfilter.h:128-..., postfilter.cc:502-945): a Zelinski object with a noise coherence matrix per bin
dsr_status dsr_mccowan_create(int fftLen, int chanN, double alpha, int type, int minFrames, float threshold, dsr_zelinski** out)
{
  const dsr_status s = dsr_zelinski_create(fftLen, chanN, alpha, type, minFrames, out);
  if (s != DSR_OK) return s;
  (*out)->kind = 1; (*out)->threshold = (double) threshold; (*out)->h_R.assign((size_t) (fftLen / 2 + 1) * chanN * chanN * 2, 0.0);
  return DSR_OK;
}
// LefkimmiatisPostFilter (postfilter.h:180-202, postfilter.cc:948-1210): a McCowan object with minSV / fbinX1; the per-bin
// d^H pinv(R) d is recomputed on the host whenever the coherence matrices or the manifold change
dsr_status dsr_lefkimmiatis_create(int fftLen, int chanN, double minSV, int fbinX1, double alpha, int type, int minFrames, float threshold, dsr_zelinski** out)
{
  const dsr_status s0 = dsr_mccowan_create(fftLen, chanN, alpha, type, minFrames, threshold, out);
  if (s0 != DSR_OK) return s0;
  (*out)->kind = 2; (*out)->minSV = minSV; (*out)->fbinX1 = fbinX1 < 0 ? 0 : fbinX1;
  return DSR_OK;
}
dsr_status dsr_mccowan_set_noise_matrix(dsr_zelinski* p, int fbinX, const double* Rnn)
{
  return guard([&] {
    if (!p || !Rnn || p->kind < 1) throw Error(DSR_E_PARAMETER, "not a McCowan post-filter");
    if (fbinX < 0 || fbinX > p->M / 2) throw Error(DSR_E_DIMENSION, "fbinX %d out of range", fbinX);
    memcpy(&p->h_R[(size_t) fbinX * p->C * p->C * 2], Rnn, sizeof(double) * 2 * p->C * p->C); p->haveR = true; p->dirtyR = true;
  });
}
dsr_status dsr_mccowan_set_diffuse_noise_model(dsr_zelinski* p, const double* micPos, double sampleRate, double sspeed)
{
  return guard([&] {                                            // postfilter.cc:568-626
    if (!p || !micPos || p->kind < 1) throw Error(DSR_E_PARAMETER, "not a McCowan post-filter");
    const int C = p->C, F = p->M / 2 + 1;
    for (int f = 0; f < F; f++) {
      const double omega_d_c = 2.0 * sampleRate * f / (p->M * sspeed);
      double* Rf = &p->h_R[(size_t) f * C * C * 2];
      for (int m = 0; m < C; m++)
        for (int n = 0; n < m; n++) {
          const double dx = micPos[3 * m] - micPos[3 * n], dy = micPos[3 * m + 1] - micPos[3 * n + 1], dz = micPos[3 * m + 2] - micPos[3 * n + 2];
          const double x = omega_d_c * sqrt(dx * dx + dy * dy + dz * dz);
          Rf[(m * C + n) * 2] = (x == 0.0) ? 1.0 : sin(M_PI * x) / (M_PI * x); Rf[(m * C + n) * 2 + 1] = 0.0;     // gsl_sf_sinc
        }
      for (int m = 0; m < C; m++) { Rf[(m * C + m) * 2] = 1.0; Rf[(m * C + m) * 2 + 1] = 0.0; }
      for (int m = 0; m < C; m++) for (int n = m + 1; n < C; n++) { Rf[(m * C + n) * 2] = Rf[(n * C + m) * 2]; Rf[(m * C + n) * 2 + 1] = Rf[(n * C + m) * 2 + 1]; }
    }
    p->haveR = true; p->dirtyR = true;
  });
}
dsr_status dsr_mccowan_diagonal_loading(dsr_zelinski* p, int fbinX, float diagonalWeight)
{
  return guard([&] {                                            // setAllLevelsOfDiagonalLoading (fbinX < 0) / setLevelOfDiagonalLoading, :628-657
    if (!p || p->kind < 1) throw Error(DSR_E_PARAMETER, "not a McCowan post-filter");
    if (!p->haveR) throw Error(DSR_E_ERROR, "Construct/set first a noise coherence matrix");
    const int C = p->C, F = p->M / 2 + 1;
    for (int f = (fbinX < 0 ? 0 : fbinX); f < (fbinX < 0 ? F : fbinX + 1); f++) for (int c = 0; c < C; c++) p->h_R[((size_t) f * C * C + c * C + c) * 2] += (double) diagonalWeight;
    p->dirtyR = true;
  });
}
dsr_status dsr_mccowan_divide_nondiagonal(dsr_zelinski* p, float myu)
{
  return guard([&] {                                            // divideAllNonDiagonalElements, :664-682 (gsl_complex_div by (1 + myu, 0))
    if (!p || p->kind < 1) throw Error(DSR_E_PARAMETER, "not a McCowan post-filter");
    const int C = p->C, F = p->M / 2 + 1; const double br = 1.0 + (double) myu;
    for (int f = 0; f < F; f++) for (int a = 0; a < C; a++) for (int b = 0; b < C; b++) if (a != b) {
      double* z = &p->h_R[((size_t) f * C * C + a * C + b) * 2];
      const double s = 1.0 / hypot(br, 0.0), sbr = s * br, sbi = s * 0.0; const double zr = (z[0] * sbr + z[1] * sbi) * s, zi = (z[1] * sbr - z[0] * sbi) * s;
      z[0] = zr; z[1] = zi;
    }
    p->dirtyR = true;
  });
}
dsr_status dsr_zelinski_set_manifold(dsr_zelinski* p, int fbinX, const double* vec)
{
  return guard([&] {
    if (!p || !vec) throw Error(DSR_E_PARAMETER, "null argument");
    if (fbinX < 0 || fbinX >= p->M) throw Error(DSR_E_DIMENSION, "fbinX %d must be less than %d", fbinX, p->M);       // :398-401
    if (fbinX > p->M / 2) return;                                  // the mirror half is implied (halfBandShift == false)
    memcpy(&p->h_wq[(size_t) fbinX * p->C * 2], vec, sizeof(double) * 2 * p->C); p->dirty = true;
  });
}
namespace dsr { const float2* bf_fixed_weights_dev(dsr_bf* s); }
// bfW: null -- Y is the beamformer's output; else the channel-major weights of the beamformer (pitch F + 1): the Zelinski streaming pass forms Y itself (into Yscr, or the
// caller's array when it wants the beamformer's output too)
static void zelinski_apply_impl(dsr_zelinski* p, const float* X, const float* Y, const float2* bfW, float* Ykeep, const int32_t* nframes_dev, int U, int Tmax, float* out, float* wp1, void* stream)
{
  {
    if (U <= 0 || Tmax <= 0) return;
    hipStream_t st = (hipStream_t) stream;
    const int F = p->M / 2 + 1, C = p->C;
    if (p->dirty) {
      std::vector<double2> w((size_t) F * C), wt((size_t) F * C);
      for (size_t i = 0; i < w.size(); i++) w[i] = make_double2(p->h_wq[2 * i], p->h_wq[2 * i + 1]);
      for (int f = 0; f < F; f++) for (int c = 0; c < C; c++) wt[(size_t) c * F + f] = w[(size_t) f * C + c];
      p->wq.upload(w); p->wqT.upload(wt); p->dirty = false; p->dirtyL = true;
    }
    const size_t S = (size_t) U * F, NE = (size_t) C * (C + 1) / 2;
    // Zelinski: the register kernel for the small arrays it is instantiated for, the two streaming kernels for every other size (and on request)
    const bool zsum = p->kind == 0 && (!(C == 2 || C == 3 || C == 4 || C == 6 || C == 8) || getenv("DSR_PF_SUM"));
    // carried state: densities [entry][U x F] + frames seen per stream (double buffered: a call reads one array and writes the other)
    int carryIn = 0, carryOut = 0; const int* seenIn = nullptr; int* seenOut = nullptr;
    if (p->carry) {
      if (p->haveState && p->stateU != U) throw Error(DSR_E_CONSISTENCY, "post-filter: the carried state holds %d streams, this call has %d (reset the state first)", p->stateU, U);
      p->state.reserve(zsum ? 2 * S : S * NE); p->seen[0].reserve(U); p->seen[1].reserve(U);
      carryIn = p->haveState ? 1 : 0; carryOut = 1;
      seenIn = p->haveState ? p->seen[p->seenCur].p : nullptr; seenOut = p->seen[p->seenCur ^ 1].p;
    }
    const bool wave = !zsum && (C > 16 || getenv("DSR_PF_WAVE"));
    if (wave) {
      if (!p->pairIJ.p) { std::vector<unsigned short> t; for (int i = 0; i < C - 1; i++) for (int j = i + 1; j < C; j++) t.push_back((unsigned short) (i | (j << 8))); p->pairIJ.upload(t); }
      if (!p->carry) p->state.reserve(16);
    }
    if (p->kind >= 1) {
      if (!p->haveR) throw Error(DSR_E_ERROR, "McCowanPostFilter: construct/set a noise coherence matrix");             // postfilter.cc:835-838
      const bool newR = p->dirtyR;
      if (p->dirtyR) { std::vector<double2> r((size_t) F * C * C); for (size_t i = 0; i < r.size(); i++) r[i] = make_double2(p->h_R[2 * i], p->h_R[2 * i + 1]); p->R.upload(r); p->dirtyR = false; }
      if (p->kind == 2 && (newR || p->dirtyL)) {                                 // calcInverseNoiseSpatialSpectralMatrix + calcLambda (postfilter.cc:981-1009)
        std::vector<double2> lam(F);
        for (int f = 0; f < F; f++) lam[f] = lefkimmiatis_lambda(&p->h_R[(size_t) f * C * C * 2], &p->h_wq[(size_t) f * C * 2], C, p->minSV);
        p->lambda.upload(lam); p->dirtyL = false;
      }
    }
#define PF_TAIL seenIn, seenOut, carryIn, carryOut
    if (zsum) {
      if (!p->carry) p->state.reserve(16);
      ZelinskiPlan::PE& pe = p->pe.at(st); const size_t TF = (size_t) Tmax * F;
      pe.P.reserve((size_t) U * TF); pe.E.reserve((size_t) U * TF);
      if (bfW) {
        float2* Yw = (float2*) Ykeep; if (!Yw) { pe.Y.reserve((size_t) U * TF); Yw = pe.Y.p; }
        hipLaunchKernelGGL(k_zel_pairs<true>, dim3((unsigned) (((size_t) ((Tmax + 1) / 2) * F + 255) / 256), (unsigned) U), dim3(256), 0, st, (const float2*) X, nframes_dev, p->wqT.p, pe.P.p, pe.E.p, C, Tmax, F, bfW, F + 1, Yw);
        Y = (const float*) Yw;
      } else
      hipLaunchKernelGGL(k_zel_pairs<false>, dim3((unsigned) (((size_t) ((Tmax + 1) / 2) * F + 255) / 256), (unsigned) U), dim3(256), 0, st, (const float2*) X, nframes_dev, p->wqT.p, pe.P.p, pe.E.p, C, Tmax, F, (const float2*) nullptr, 0, (float2*) nullptr);
      hipLaunchKernelGGL(k_zel_recur, dim3((unsigned) ((S + 63) / 64)), dim3(1024), 0, st, pe.P.p, pe.E.p, (const float2*) Y, nframes_dev, p->state.p, (float2*) out, wp1,
                         U, C, Tmax, F, p->alpha, p->type, p->minFrames, PF_TAIL);
    } else if (wave) {
      const size_t ldsPf = p->kind ? sizeof(double2) * (size_t) C * (C - 1) / 2 : 0;
#define PFW(K) hipLaunchKernelGGL((k_pf_wave<K>), dim3((unsigned) S), dim3(64), ldsPf, st, (const float2*) X, (const float2*) Y, nframes_dev, p->wq.p, p->R.p, p->pairIJ.p, p->state.p, \
                                  (float2*) out, wp1, U, C, Tmax, F, p->alpha, p->type, p->minFrames, p->threshold, p->kind == 2 ? p->lambda.p : nullptr, p->fbinX1, PF_TAIL)
      if (p->kind == 0) PFW(0); else if (p->kind == 1) PFW(1); else PFW(2);
#undef PFW
    } else if (p->kind >= 1) {
      const bool regsM = !getenv("DSR_PF_MEMSTATE") && (C == 2 || C == 3 || C == 4 || C == 6 || C == 8);
      if (!p->carry) p->state.reserve(regsM ? 16 : S * NE);
      // (state in memory: the working array is the carried one, in place -- the recursions restart by themselves on a stream's first two frames)
#define MC_LAUNCH(CTV) hipLaunchKernelGGL((k_mccowan<CTV>), dim3((unsigned) ((S + 127) / 128)), dim3(128), 0, st, (const float2*) X, (const float2*) Y, nframes_dev, p->wq.p, p->R.p, p->state.p, \
                         (float2*) out, wp1, U, C, Tmax, F, p->alpha, p->type, p->minFrames, p->threshold, p->kind == 2 ? p->lambda.p : nullptr, p->fbinX1, PF_TAIL);
      if (!regsM) { MC_LAUNCH(0) } else if (C == 8) { MC_LAUNCH(8) } else if (C == 6) { MC_LAUNCH(6) } else if (C == 4) { MC_LAUNCH(4) } else if (C == 3) { MC_LAUNCH(3) } else { MC_LAUNCH(2) }
#undef MC_LAUNCH
    } else {
      if (!p->carry) p->state.reserve(16);
#define ZREG(CC) if (C == CC) hipLaunchKernelGGL(k_zelinski_reg<CC>, dim3((unsigned) ((S + 63) / 64)), dim3(64), 0, st, (const float2*) X, (const float2*) Y, \
      nframes_dev, p->wq.p, (float2*) out, wp1, U, Tmax, F, p->alpha, p->type, p->minFrames, seenIn, seenOut, p->state.p, carryIn, carryOut);
      ZREG(2) ZREG(3) ZREG(4) ZREG(6) ZREG(8)
#undef ZREG
    }
#undef PF_TAIL
    DSR_HIP(hipGetLastError());
    if (p->carry) { p->haveState = true; p->stateU = U; p->seenCur ^= 1; }
  }
}
dsr_status dsr_zelinski_apply(dsr_zelinski* p, const float* X, const float* Y, const int32_t* nframes_dev, int U, int Tmax, float* out, float* wp1, void* stream)
{
  return guard([&] {
    if (!p || !X || !Y || !nframes_dev || !out) throw Error(DSR_E_PARAMETER, "null argument");
    zelinski_apply_impl(p, X, Y, nullptr, nullptr, nframes_dev, U, Tmax, out, wp1, stream);
  });
}
// The post-filter behind its beamformer (ZelinskiPostFilter::setBeamformer, postfilter.cc:376-384: the filter takes snapshots and array manifold from the beamformer
// whose output it filters): out = postfilter(X, bf(X)).  Where the filter streams the snapshots anyway (Zelinski on arrays the register kernel is not instantiated
// for) the beamformer's sum is formed in the same pass; elsewhere this is dsr_bf_apply_frames followed by dsr_zelinski_apply.  Y_dev (optional) receives bf(X).
dsr_status dsr_zelinski_apply_bf(dsr_zelinski* p, dsr_bf* bf, const float* X, const int32_t* nframes_dev, int U, int Tmax, float* out, float* wp1, float* Y_dev, void* stream)
{
  return guard([&] {
    if (!p || !bf || !X || !nframes_dev || !out) throw Error(DSR_E_PARAMETER, "null argument");
    if (U <= 0 || Tmax <= 0) return;
    const int F = p->M / 2 + 1, C = p->C;
    const bool zsum = p->kind == 0 && (!(C == 2 || C == 3 || C == 4 || C == 6 || C == 8) || getenv("DSR_PF_SUM"));
    const float2* w = (zsum && !getenv("DSR_PF_NOFUSE")) ? bf_fixed_weights_dev(bf) : nullptr;
    if (w) { zelinski_apply_impl(p, X, nullptr, w, Y_dev, nframes_dev, U, Tmax, out, wp1, stream); return; }
    float* Y = Y_dev;
    if (!Y) { ZelinskiPlan::PE& pe = p->pe.at((hipStream_t) stream); pe.Y.reserve((size_t) U * Tmax * F); Y = (float*) pe.Y.p; }
    { const dsr_status rc = dsr_bf_apply_frames(bf, X, nframes_dev, U, Tmax, Y, stream); if (rc != DSR_OK) throw Error(rc, "%s", dsr_last_error()); }
    zelinski_apply_impl(p, X, Y, nullptr, nullptr, nframes_dev, U, Tmax, out, wp1, stream);
  });
}
// Carried densities (block streaming): carry = 1 makes every apply of the same U continue the recursions of the call before it (stream u of one call
// = stream u of the next; alpha = 0 only on a stream's own first two frames, minFrames counted from its start); reset_state starts new streams.
dsr_status dsr_zelinski_carry(dsr_zelinski* p, int on)
{ return guard([&] { if (!p) throw Error(DSR_E_PARAMETER, "null argument"); p->carry = on != 0; if (!on) p->haveState = false; }); }
dsr_status dsr_zelinski_reset_state(dsr_zelinski* p)
{ return guard([&] { if (!p) throw Error(DSR_E_PARAMETER, "null argument"); p->haveState = false; }); }

}  // extern "C"
