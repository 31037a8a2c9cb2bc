// csrc/k_wpe.hip -- single-channel weighted-prediction-error (WPE) dereverberation of subband sequences
// (second operator).
//
// Replaces SingleChannelWPEDereverberationFeature (btk/dereverberation/dereverberation.cc:28-300): _getLags, _calculateThetan,
// _calculateRr, _loadR, _estimateGn (gsl_linalg_complex_cholesky_decomp/_solve) and next().
//
// One workgroup owns one (utterance, subband): the subband's whole time series sits in LDS (fp64), the P x P weighted
// correlation matrix is accumulated with one thread per lower-triangle entry (frames in order), the P-tap prediction filter
// comes from an in-LDS Cholesky factorisation.  Everything is fp64 like the reference.  Deviation kept on purpose: the terms
// are weighted with the reciprocal of theta_n (one division per frame instead of one per term); agreement with the oracle is
// 1e-9 relative.  The filters start from zero for every utterance (the reference's reset() keeps them from the previous
// one; nextSpeaker() zeroes them, :283-290).
#include "common.h"
#include <cmath>

namespace dsr {

__global__ __launch_bounds__(256) void k_wpe(const float2* __restrict__ Y, const int* __restrict__ nframesArr, float2* __restrict__ out,
                                             double2* gnOut, int U, int Nmax, int F, int M, int lowerN, int P, int iterationsN,
                                             double loadFactor, int lowerBW, const double2* gnIn)      // (gnIn and gnOut may be ONE buffer -- dsr_wpe_single_continue: no __restrict__)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* y = reinterpret_cast<double2*>(smem);                 // [N]
  double* rth = reinterpret_cast<double*>(y + Nmax);             // [N]  1 / theta_n
  double2* R = reinterpret_cast<double2*>(rth + Nmax);           // [P][P] lower triangle
  double2* r = R + P * P;                                        // [P]
  double2* g = r + P;                                            // [P]
  __shared__ int s_fail;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int b = blockIdx.x, u = blockIdx.y;
  const int N = nframesArr[u] < Nmax ? nframesArr[u] : Nmax;
  const float2* Yu = Y + (long) u * Nmax * F; float2* Ou = out + (long) u * Nmax * F;
  const bool selected = (b <= lowerBW) || (b >= M - lowerBW);    // dereverberation.cc:204,241
  for (int n = tid; n < N; n += nthr) { const float2 v = Yu[(long) n * F + b]; y[n] = make_double2((double) v.x, (double) v.y); }
  for (int l = tid; l < P; l += nthr) g[l] = gnIn ? gnIn[((long) u * F + b) * P + l] : make_double2(0.0, 0.0);   // reset() keeps _gn, nextSpeaker() zeroes it (:258-277)
  if (tid == 0) s_fail = 0;
  __syncthreads();
  auto predict = [&](int n) -> double2 {                         // zdotc(gn, lags(n - lowerN)) = sum_l conj(g_l) y[n - lowerN - l]
    double dr = 0.0, di = 0.0;
    for (int l = 0; l < P; l++) {
      const int ix = n - lowerN - l; if (ix < 0) break;
      const double gr = g[l].x, gi = -g[l].y; const double2 v = y[ix];
      dr += gr * v.x - gi * v.y; di += gr * v.y + gi * v.x;
    }
    return make_double2(dr, di);
  };
  if (selected) {
    for (int it = 0; it < iterationsN; it++) {
      for (int n = tid; n < N; n += nthr) {                      // _calculateThetan
        double2 c = y[n];
        if (n >= lowerN) { const double2 d = predict(n); c.x -= d.x; c.y -= d.y; }
        double th = hypot(c.x, c.y); if (th < 1.0E-03) th = 1.0E-03;
        rth[n] = 1.0 / (th * th);
      }
      __syncthreads();
      // _calculateRr: lower triangle of R, then r.  LPE lanes share an entry's sum over the frames (a power of two that divides the wave: the partial sums meet
      // by lane exchange) -- with 4 taps that is 14 entries x 16 lanes instead of 14 threads walking 1250 frames each while 242 wait (1.6 -> 0.3 ms at 32 streams x 129 bins).
      const int nEnt = P * (P + 1) / 2;
      int LPE = 1; while (LPE < 64 && (nEnt + P) * LPE * 2 <= nthr) LPE *= 2;
      const int grp = tid / LPE, lig = tid % LPE, nGrp = nthr / LPE;
      for (int e0 = 0; e0 < nEnt + P; e0 += nGrp) {
        const int e = e0 + grp; const bool on = e < nEnt + P;
        double sr = 0.0, si = 0.0; int row = 0, col = 0;
        if (on && e < nEnt) {
          row = (int) ((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while (row * (row + 1) / 2 > e) row--; while ((row + 1) * (row + 2) / 2 <= e) row++;
          col = e - row * (row + 1) / 2;
          for (int n = lowerN + row + lig; n < N; n += LPE) {    // lag[row] = y[n - lowerN - row] (zero before the start)
            const double2 a = y[n - lowerN - row], c = y[n - lowerN - col]; const double w = rth[n];
            sr += (a.x * c.x + a.y * c.y) * w; si += (a.y * c.x - a.x * c.y) * w;      // a conj(c)
          }
        } else if (on) {
          const int l = e - nEnt;
          for (int n = lowerN + l + lig; n < N; n += LPE) {
            const double2 c = y[n], a = y[n - lowerN - l]; const double w = rth[n];
            sr += (c.x * a.x + c.y * a.y) * w; si += (c.x * a.y - c.y * a.x) * w;      // conj(current) lag_l
          }
        }
        for (int d = LPE >> 1; d > 0; d >>= 1) { sr += __shfl_xor(sr, d, 64); si += __shfl_xor(si, d, 64); }
        if (on && lig == 0) { if (e < nEnt) R[row * P + col] = make_double2(sr, si); else r[e - nEnt] = make_double2(sr, si); }
      }
      __syncthreads();
      if (tid == 0) {                                            // _loadR, Cholesky (lower), two triangular solves
        double maxd = 0.0;
        for (int c = 0; c < P; c++) { const double d = hypot(R[c * P + c].x, R[c * P + c].y); if (d > maxd) maxd = d; }
        for (int c = 0; c < P; c++) { const double d = hypot(R[c * P + c].x, R[c * P + c].y) + maxd * loadFactor; R[c * P + c] = make_double2(d, 0.0); }
        bool ok = true;
        for (int j = 0; j < P && ok; j++) {
          double ajj = R[j * P + j].x;
          for (int k = 0; k < j; k++) ajj -= R[j * P + k].x * R[j * P + k].x + R[j * P + k].y * R[j * P + k].y;
          if (ajj <= 0.0) { ok = false; break; }
          ajj = sqrt(ajj); R[j * P + j] = make_double2(ajj, 0.0);
          for (int i = j + 1; i < P; i++) {
            double sr = R[i * P + j].x, si = R[i * P + j].y;
            for (int k = 0; k < j; k++) { const double2 a = R[i * P + k], c = R[j * P + k]; sr -= a.x * c.x + a.y * c.y; si -= a.y * c.x - a.x * c.y; }
            R[i * P + j] = make_double2(sr / ajj, si / ajj);
          }
        }
        if (!ok) s_fail = 1;
        else {
          for (int i = 0; i < P; i++) {
            double sr = r[i].x, si = r[i].y;
            for (int k = 0; k < i; k++) { const double2 a = R[i * P + k]; sr -= a.x * g[k].x - a.y * g[k].y; si -= a.x * g[k].y + a.y * g[k].x; }
            const double d = R[i * P + i].x; g[i] = make_double2(sr / d, si / d);
          }
          for (int i = P - 1; i >= 0; i--) {
            double sr = g[i].x, si = g[i].y;
            for (int k = i + 1; k < P; k++) { const double ar = R[k * P + i].x, ai = -R[k * P + i].y; sr -= ar * g[k].x - ai * g[k].y; si -= ar * g[k].y + ai * g[k].x; }
            const double d = R[i * P + i].x; g[i] = make_double2(sr / d, si / d);
          }
        }
      }
      __syncthreads();
      if (s_fail) break;
    }
  }
  const bool fail = s_fail != 0;
  for (int n = tid; n < Nmax; n += nthr) {                       // next(): subtract the predicted late reverberation
    float2 o = make_float2(0.f, 0.f);
    if (n < N) {
      double2 c = y[n];
      if (selected && n >= lowerN) { const double2 d = predict(n); c.x -= d.x; c.y -= d.y; }
      o = fail ? make_float2(NAN, NAN) : make_float2((float) c.x, (float) c.y);
    }
    Ou[(long) n * F + b] = o;
  }
  if (gnOut) for (int l = tid; l < P; l += nthr) gnOut[((long) u * F + b) * P + l] = g[l];
}


// MultiChannelWPEDereverberation (dereverberation.cc:281-586).  One workgroup owns one (utterance, subband, channel): theta_n, the stacked lag
// vector [channel][lag] (_getLags :422-437), the (C P) x (C P) weighted correlation matrix, its loading, Cholesky and the prediction filter are
// this channel's (:439-573).  The matrix is Hermitian: only its lower triangle is kept, packed (entry (i, j), j <= i, at i (i + 1) / 2 + j) -- half
// the LDS of the square, which is what lets 8 channels x 10 taps (80 x 80) sit next to the subband's series of all channels at the benchmark's
// 1257 frames.  YLDS: the series (fp32 as delivered, widened when used) are staged in LDS; otherwise they are read from a copy of the snapshots
// transposed to [utterance][subband][channel][frame] (k_wpe_series: contiguous per subband, so the accumulation loops hit L1/L2 lines whole) and
// only the matrix, the filters and 1 / theta_n live in LDS.  The filters go to memory; k_wpe_multi_out then subtracts the predictions (getOutput :365-395).
__global__ __launch_bounds__(256) void k_wpe_series(const float2* __restrict__ Y, float2* __restrict__ Yt, int C, int Nmax, int F)
{
  // Yt[((u F + b) C + ch) Nmax + n] = Y[((u C + ch) Nmax + n) F + b]: a 32 x 32 (frame, bin) tile per workgroup through LDS, both sides coalesced
  __shared__ float2 tile[32][33];
  const int uc = blockIdx.z, u = uc / C, ch = uc - u * C;
  const int n0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) { const int n = n0 + j, b = b0 + tx; if (n < Nmax && b < F) tile[j][tx] = Y[((long) uc * Nmax + n) * F + b]; }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) { const int b = b0 + j, n = n0 + tx; if (n < Nmax && b < F) Yt[(((long) u * F + b) * C + ch) * Nmax + n] = tile[tx][j]; }
}

template <bool YLDS>
__global__ __launch_bounds__(256) void k_wpe_multi(const float2* __restrict__ Y, const float2* __restrict__ Yt, const int* __restrict__ nframesArr,
                                                   double2* __restrict__ gnOut, int U, int C, int Nmax, int F, int M, int lowerN, int P, int iterationsN,
                                                   double loadFactor, int lowerBW)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int PT = P * C, nEnt = PT * (PT + 1) / 2;
  double2* R = reinterpret_cast<double2*>(smem);                 // packed lower triangle
  double2* r = R + nEnt;                                         // [PT]
  double2* g = r + PT;                                           // [PT]
  double* rth = reinterpret_cast<double*>(g + PT);               // [N]  1 / theta_n
  float2* yl = reinterpret_cast<float2*>(rth + Nmax);            // [C][N] (YLDS)
  __shared__ int s_fail;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int b = blockIdx.x, c0 = blockIdx.y, u = blockIdx.z;
  const int N = nframesArr[u] < Nmax ? nframesArr[u] : Nmax;
  const bool selected = (b <= lowerBW) || (b >= M - lowerBW);    // dereverberation.cc:552
  double2* gOut = gnOut + (((long) u * C + c0) * F + b) * PT;
  if (!selected) { for (int l = tid; l < PT; l += nthr) gOut[l] = make_double2(0.0, 0.0); return; }
  if (YLDS) for (int i = tid; i < C * N; i += nthr) { const int ch = i / N, n = i - ch * N; yl[ch * Nmax + n] = Y[(((long) u * C + ch) * Nmax + n) * F + b]; }
  const float2* y = YLDS ? yl : Yt + ((long) u * F + b) * C * Nmax;      // [C][Nmax] either way (two instantiations: no pointer select at run time)
  auto tri = [](int i, int j) { return i * (i + 1) / 2 + j; };
  for (int l = tid; l < PT; l += nthr) g[l] = make_double2(0.0, 0.0);
  if (tid == 0) s_fail = 0;
  __syncthreads();
  for (int it = 0; it < iterationsN; it++) {
    for (int n = tid; n < N; n += nthr) {                        // _calculateThetan
      const float2 v = y[c0 * Nmax + n]; double cr = (double) v.x, ci = (double) v.y;
      if (n >= lowerN) {
        double dr = 0.0, di = 0.0;
        for (int ch = 0; ch < C; ch++)
          for (int l = 0; l < P; l++) {                             // t = ch * P + l, in the reference's order of the stacked lags
            const int ix = n - lowerN - l; if (ix < 0) break;
            const float2 v = y[ch * Nmax + ix]; const double gr = g[ch * P + l].x, gi = -g[ch * P + l].y;
            dr += gr * (double) v.x - gi * (double) v.y; di += gr * (double) v.y + gi * (double) v.x;
          }
        cr -= dr; ci -= di;
      }
      double th = hypot(cr, ci); if (th < 1.0E-03) th = 1.0E-03;
      rth[n] = 1.0 / (th * th);
    }
    __syncthreads();
    for (int e = tid; e < nEnt + PT; e += nthr) {                // _calculateRr: lower triangle of R, then r
      double sr = 0.0, si = 0.0;
      if (e < nEnt) {
        int row = (int) ((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while (row * (row + 1) / 2 > e) row--; while ((row + 1) * (row + 2) / 2 <= e) row++;
        const int col = e - row * (row + 1) / 2;
        const int chR = row / P, lR = row - chR * P, chC = col / P, lC = col - chC * P;
        const float2* yr = y + chR * Nmax - lowerN - lR; const float2* yc = y + chC * Nmax - lowerN - lC;
        for (int n = lowerN + (lR > lC ? lR : lC); n < N; n++) {    // a lag before the start of the utterance is zero: those frames add nothing
          const float2 a = yr[n], q = yc[n]; const double w = rth[n];
          sr += ((double) a.x * (double) q.x + (double) a.y * (double) q.y) * w; si += ((double) a.y * (double) q.x - (double) a.x * (double) q.y) * w;
        }
        R[e] = make_double2(sr, si);                             // (e IS the packed index of (row, col))
      } else {
        const int l = e - nEnt;
        const int chL = l / P, lL = l - chL * P; const float2* ylag = y + chL * Nmax - lowerN - lL;
        for (int n = lowerN + lL; n < N; n++) { const float2 v = y[c0 * Nmax + n], a = ylag[n]; const double w = rth[n];
          sr += ((double) v.x * (double) a.x + (double) v.y * (double) a.y) * w; si += ((double) v.x * (double) a.y - (double) v.y * (double) a.x) * w; }
        r[l] = make_double2(sr, si);
      }
    }
    __syncthreads();
    if (tid == 0) {                                              // _loadR
      double maxd = 0.0;
      for (int k = 0; k < PT; k++) { const double2 v = R[tri(k, k)]; const double d = hypot(v.x, v.y); if (d > maxd) maxd = d; }
      for (int k = 0; k < PT; k++) { const double2 v = R[tri(k, k)]; const double d = hypot(v.x, v.y) + maxd * loadFactor; R[tri(k, k)] = make_double2(d, 0.0); }
    }
    __syncthreads();
    for (int j = 0; j < PT; j++) {                               // Cholesky (lower): the diagonal entry by one thread, the column below it by all
      if (tid == 0) {
        double ajj = R[tri(j, j)].x;
        for (int k = 0; k < j; k++) { const double2 v = R[tri(j, k)]; ajj -= v.x * v.x + v.y * v.y; }
        if (ajj <= 0.0) s_fail = 1; else R[tri(j, j)] = make_double2(sqrt(ajj), 0.0);
      }
      __syncthreads();
      if (s_fail) break;
      const double ajj = R[tri(j, j)].x;
      for (int i = j + 1 + tid; i < PT; i += nthr) {
        double sr = R[tri(i, j)].x, si = R[tri(i, j)].y;
        for (int k = 0; k < j; k++) { const double2 a = R[tri(i, k)], q = R[tri(j, k)]; sr -= a.x * q.x + a.y * q.y; si -= a.y * q.x - a.x * q.y; }
        R[tri(i, j)] = make_double2(sr / ajj, si / ajj);
      }
      __syncthreads();
    }
    if (tid == 0) {                                              // two triangular solves
      if (!s_fail) {
        for (int i = 0; i < PT; i++) {
          double sr = r[i].x, si = r[i].y;
          for (int k = 0; k < i; k++) { const double2 a = R[tri(i, k)]; sr -= a.x * g[k].x - a.y * g[k].y; si -= a.x * g[k].y + a.y * g[k].x; }
          const double d = R[tri(i, i)].x; g[i] = make_double2(sr / d, si / d);
        }
        for (int i = PT - 1; i >= 0; i--) {
          double sr = g[i].x, si = g[i].y;
          for (int k = i + 1; k < PT; k++) { const double ar = R[tri(k, i)].x, ai = -R[tri(k, i)].y; sr -= ar * g[k].x - ai * g[k].y; si -= ar * g[k].y + ai * g[k].x; }
          const double d = R[tri(i, i)].x; g[i] = make_double2(sr / d, si / d);
        }
      }
    }
    __syncthreads();
    if (s_fail) break;
  }
  for (int l = tid; l < PT; l += nthr) gOut[l] = s_fail ? make_double2(NAN, NAN) : g[l];
}

// getOutput: one thread per (utterance, channel, frame, bin); filterChan >= 0: every channel through that channel's filter
__global__ __launch_bounds__(256) void k_wpe_multi_out(const float2* __restrict__ Y, const int* __restrict__ nframesArr, const double2* __restrict__ gn,
                                                       float2* __restrict__ out, int U, int C, int Nmax, int F, int M, int lowerN, int P, int lowerBW, int filterChan)
{
  const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long) U * C * Nmax * F) return;
  const int b = (int) (i % F); long q = i / F; const int n = (int) (q % Nmax); q /= Nmax; const int c = (int) (q % C), u = (int) (q / C);
  const int N = nframesArr[u] < Nmax ? nframesArr[u] : Nmax;
  if (n >= N) { out[i] = make_float2(0.f, 0.f); return; }
  const float2 v = Y[i]; double cr = (double) v.x, ci = (double) v.y;
  if (n >= lowerN && ((b <= lowerBW) || (b >= M - lowerBW))) {
    const int fc = filterChan >= 0 ? filterChan : c, PT = P * C;
    const double2* g = gn + (((long) u * C + fc) * F + b) * PT;
    double dr = 0.0, di = 0.0;
    for (int ch = 0; ch < C; ch++)
      for (int l = 0; l < P; l++) {
        const int ix = n - lowerN - l; if (ix < 0) break;
        const float2 a = Y[(((long) u * C + ch) * Nmax + ix) * F + b]; const double2 gg = g[ch * P + l];
        dr += gg.x * (double) a.x + gg.y * (double) a.y; di += gg.x * (double) a.y - gg.y * (double) a.x;      // conj(g) a
      }
    cr -= dr; ci -= di;
  }
  out[i] = make_float2((float) cr, (float) ci);
}

}  // namespace dsr

using namespace dsr;

extern "C" {

// Y_dev [U][Nmax][M/2+1] complex64 (one channel's subband snapshots or a beamformer output), nframes_dev [U] -> out_dev same shape;
// gn_dev (optional) [U][M/2+1][P] complex128 prediction filters.  A subband whose loaded correlation matrix is not positive definite
// (GSL would abort there) yields NaNs.
static dsr_status wpe_single_impl(const float* Y_dev, const int32_t* nframes_dev, int U, int Nmax, int fftLen, int lowerN, int upperN, int iterationsN,
                                  double loadDb, double bandWidth, double sampleRate, float* out_dev, double* gn_dev, bool warm, void* stream);
dsr_status dsr_wpe_single(const float* Y_dev, const int32_t* nframes_dev, int U, int Nmax, int fftLen, int lowerN, int upperN, int iterationsN,
                          double loadDb, double bandWidth, double sampleRate, float* out_dev, double* gn_dev, void* stream)
{ return wpe_single_impl(Y_dev, nframes_dev, U, Nmax, fftLen, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate, out_dev, gn_dev, false, stream); }
// the same for the next utterance (or block of a long stream) of an object that was reset() but not nextSpeaker()-ed: gn_dev (required) holds the
// filters the utterance before left; they seed the first theta_n (dereverberation.cc:152-176) and are replaced by this utterance's
dsr_status dsr_wpe_single_continue(const float* Y_dev, const int32_t* nframes_dev, int U, int Nmax, int fftLen, int lowerN, int upperN, int iterationsN,
                                   double loadDb, double bandWidth, double sampleRate, float* out_dev, double* gn_dev, void* stream)
{ return wpe_single_impl(Y_dev, nframes_dev, U, Nmax, fftLen, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate, out_dev, gn_dev, true, stream); }
static dsr_status wpe_single_impl(const float* Y_dev, const int32_t* nframes_dev, int U, int Nmax, int fftLen, int lowerN, int upperN, int iterationsN,
                                  double loadDb, double bandWidth, double sampleRate, float* out_dev, double* gn_dev, bool warm, void* stream)
{
  return guard([&] {
    if (!Y_dev || !nframes_dev || !out_dev || (warm && !gn_dev)) throw Error(DSR_E_PARAMETER, "null argument");
    if (upperN < lowerN || lowerN < 0 || iterationsN < 0) throw Error(DSR_E_PARAMETER, "bad prediction range [%d, %d]", lowerN, upperN);
    if (bandWidth > sampleRate / 2.0) throw Error(DSR_E_DIMENSION, "Bandwidth is greater than the Nyquist rate.");          // :261-262
    if (U <= 0 || Nmax <= 0) return;
    require_device();
    const int P = upperN - lowerN + 1, F = fftLen / 2 + 1;
    const int lowerBW = (bandWidth == 0.0) ? fftLen / 2 : (int) (unsigned) ((bandWidth / (sampleRate / 2.0)) * (fftLen / 2));
    const size_t lds = (size_t) Nmax * 24 + (size_t) (P * P + 2 * P) * 16;
    if (lds > 150 * 1024) throw Error(DSR_E_DIMENSION, "WPE: %d frames x %d taps do not fit the LDS working set", Nmax, P);
    DSR_HIP(hipFuncSetAttribute((const void*) k_wpe, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipLaunchKernelGGL(k_wpe, dim3(F, U), dim3(256), lds, (hipStream_t) stream, (const float2*) Y_dev, nframes_dev, (float2*) out_dev, (double2*) gn_dev,
                       U, Nmax, F, fftLen, lowerN, P, iterationsN, pow(10.0, loadDb / 10.0), lowerBW, warm ? (const double2*) gn_dev : nullptr);
    DSR_HIP(hipGetLastError());
  });
}


// MultiChannelWPEDereverberation (dereverberation.h:89-157, dereverberation.cc:281-586).  Y_dev [U][C][Nmax][M/2+1] complex64 -> out_dev same shape;
// gn_dev [U][C][M/2+1][C*P] complex128 (required: the filters are handed from the estimation kernel to the output kernel through it).
// filterChan < 0: every channel is filtered with its own prediction filter; >= 0: all channels with that channel's filter, which is what
// the reference's getOutput does for the channel whose feature asks for a frame first (:381).  A (subband, channel) whose loaded matrix is
// not positive definite yields NaNs.
dsr_status dsr_wpe_multi(const float* Y_dev, const int32_t* nframes_dev, int U, int chanN, int Nmax, int fftLen, int lowerN, int upperN, int iterationsN,
                         double loadDb, double bandWidth, double sampleRate, int filterChan, float* out_dev, double* gn_dev, void* stream)
{
  return guard([&] {
    if (!Y_dev || !nframes_dev || !out_dev || !gn_dev) throw Error(DSR_E_PARAMETER, "null argument");
    if (upperN < lowerN || lowerN < 0 || iterationsN < 0 || chanN < 1 || filterChan >= chanN) throw Error(DSR_E_PARAMETER, "bad prediction range [%d, %d] / channels %d / filter channel %d", lowerN, upperN, chanN, filterChan);
    if (bandWidth > sampleRate / 2.0) throw Error(DSR_E_DIMENSION, "Bandwidth is greater than the Nyquist rate.");          // :335-336
    if (U <= 0 || Nmax <= 0) return;
    require_device();
    const int P = upperN - lowerN + 1, F = fftLen / 2 + 1, PT = P * chanN;
    const int lowerBW = (bandWidth == 0.0) ? fftLen / 2 : (int) (unsigned) ((bandWidth / (sampleRate / 2.0)) * (fftLen / 2));
    // LDS: packed triangle + r + g + 1 / theta_n, and the series of all channels when they fit beside them; otherwise the series come from a
    // transposed copy of the snapshots in memory (one per stream: the copy is handed from its kernel to the estimation kernel of the same stream)
    const size_t ldsCore = ((size_t) PT * (PT + 1) / 2 + 2 * (size_t) PT) * 16 + (size_t) Nmax * 8, ldsSeries = (size_t) chanN * Nmax * 8, ldsCap = 150 * 1024;
    if (ldsCore > ldsCap) throw Error(DSR_E_DIMENSION, "WPE: %d channels x %d taps (a %d x %d matrix) and %d frames do not fit the LDS working set", chanN, P, PT, PT, Nmax);
    const bool yLds = ldsCore + ldsSeries <= ldsCap && !getenv("DSR_WPE_SERIES_MEM");
    hipStream_t st = (hipStream_t) stream;
    if (yLds) {
      const size_t lds = ldsCore + ldsSeries;
      DSR_HIP(hipFuncSetAttribute((const void*) k_wpe_multi<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      hipLaunchKernelGGL(k_wpe_multi<true>, dim3(F, chanN, U), dim3(256), lds, st, (const float2*) Y_dev, (const float2*) nullptr, nframes_dev, (double2*) gn_dev,
                         U, chanN, Nmax, F, fftLen, lowerN, P, iterationsN, pow(10.0, loadDb / 10.0), lowerBW);
    } else {
      float2* Yt = nullptr;                                          // stream-ordered scratch: lives from here to the end of the estimation kernel on this stream
      DSR_HIP(hipMallocAsync((void**) &Yt, sizeof(float2) * (size_t) U * F * chanN * Nmax, st));
      hipLaunchKernelGGL(k_wpe_series, dim3((F + 31) / 32, (Nmax + 31) / 32, U * chanN), dim3(256), 0, st, (const float2*) Y_dev, Yt, chanN, Nmax, F);
      DSR_HIP(hipFuncSetAttribute((const void*) k_wpe_multi<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsCore));
      hipLaunchKernelGGL(k_wpe_multi<false>, dim3(F, chanN, U), dim3(256), ldsCore, st, (const float2*) Y_dev, (const float2*) Yt, nframes_dev, (double2*) gn_dev,
                         U, chanN, Nmax, F, fftLen, lowerN, P, iterationsN, pow(10.0, loadDb / 10.0), lowerBW);
      DSR_HIP(hipFreeAsync(Yt, st));
    }
    DSR_HIP(hipGetLastError());
    const long tot = (long) U * chanN * Nmax * F;
    hipLaunchKernelGGL(k_wpe_multi_out, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, st, (const float2*) Y_dev, nframes_dev, (const double2*) gn_dev,
                       (float2*) out_dev, U, chanN, Nmax, F, fftLen, lowerN, P, lowerBW, filterChan);
    DSR_HIP(hipGetLastError());
  });
}

}  // extern "C"
