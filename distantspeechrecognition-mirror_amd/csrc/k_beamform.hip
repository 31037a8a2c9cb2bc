// csrc/k_beamform.hip -- subband beamformers: host-side weight design + the per-bin apply kernel.
//
// Weight design is set-up work exactly as in the reference (done once per array geometry):
//   beamformerWeights::calcMainlobe              btk/beamformer/beamformer.cc:531-594
//   SubbandMVDR::setDiffuseNoiseModel            :2486-2553      divideNonDiagonalElements  beamformer.h:362-378
//   SubbandMVDR::setAllLevelsOfDiagonalLoading   :2555-2567      setNoiseSpatialSpectralMatrix :2454-2477
//   pseudoinverse (complex<float> SVD)           :253-305        calcMVDRWeights :2392-2446
//   _calcBlockingMatrix                          :398-479        calcSidelobeCancellerP_f :761-783
// The hot loop -- SnapShotArray::update + one zdotc per bin per frame (:76-90, :2609-2631) -- is
// k_bf_apply: one thread per (frame, bin), channel spectra read coalesced along the bin axis from the
// [chan][frame][bin] layout the analysis kernel writes, weights staged in LDS.
#include "common.h"
#include "svd_linpack.h"
#include <complex>
#include <cmath>

namespace dsr {

typedef std::complex<double> zc;
typedef std::complex<float> cf;

struct BfState {
  int M = 0, C = 0, halfBandShift = 0, mode = 0;
  std::vector<zc> wq;      // [M][C]
  std::vector<zc> R;       // [M/2+1][C][C]
  bool haveR = false, haveWq = false, haveMvdr = false, haveB = false;
  std::vector<zc> mvdr;    // [M/2+1][C]
  std::vector<zc> B;       // [M][C][C-1]
  std::vector<zc> wa;      // [M][C-1]
  std::vector<zc> wl;      // [M][C]  B wa as of the last setActiveWeights_f / zeroActiveWeights (SubbandMVDRGSC reads this cached product)
  std::vector<zc> eff;     // [M/2+1][C] weights in use
  DevBuf<float2> d_w;      // [M/2+1][C]
  DevBuf<float2> d_wT;     // [C][M/2+2] the same weights channel-major (the fused analysis + beamformer kernel reads a channel's row coalesced)
  bool dirty = true;
  // SubbandGSCRLS (beamformer.h:213-262): recursive-least-squares adaptation of the active weights
  bool rlsOn = false, rlsAdapt = true, haveP0 = false; double rlsMyu = 0.9, rlsAlpha = -1.0; int rlsQc = 0;
  std::vector<double> rlsDiag;   // [M/2+1] _diagonalWeights (the constructor's sigma2)
  std::vector<zc> rlsP0;         // [M/2+1][C-1][C-1] precision matrices every utterance starts from
  DevBuf<double2> d_wq, d_B, d_P0, d_state; DevBuf<double> d_diag; bool rlsDirty = true;
  // carried adaptation state (dsr_bf_rls_carry): [entry][U x F] precision matrices + active weights as the last call left them; the next call of
  // the same batch shape continues from it -- block streaming, and the reference's "keeps adapting across reset()" (beamformer.cc:1552-1612)
  bool rlsCarry = false, rlsHaveState = false; int rlsStateU = 0; DevBuf<double2> d_carry;
  PerStream<DevBuf<float2>> d_wB;   // blockingMatrixOutput's weights, one image per stream
};

// beamformer.cc:253-305: the reference's pseudo-inverse runs LINPACK's csvdc in complex<float>; svd_linpack.cpp restates that routine
// operation for operation (pinned against the reference's own csvdc: tests/golden/linpack_csvdc.npz).  A, invA row-major n x n.  Returns
// false when a singular value fell below the threshold (the caller then uses identity, :2425-2427).
static bool pseudoinverse_cf(const zc* A, zc* invA, int n, float thr) { return linpack::pseudoinverse(A, invA, n, n, thr); }

static double sinc_pi(double x) { return std::fabs(x) < 1e-300 ? 1.0 : std::sin(M_PI * x) / (M_PI * x); }   // gsl_sf_sinc

static void calc_mainlobe(BfState& s, double fs, const double* delays)
{
  const int M = s.M, C = s.C, M2 = M / 2;
  s.wq.assign((size_t) M * C, zc(0, 0));
  if (s.halfBandShift) {                                   // beamformer.cc:544-555
    const float fshift = 0.5f;
    for (int f = 0; f < M2; f++)
      for (int c = 0; c < C; c++) {
        const double val = -2.0 * M_PI * (fshift + f) * fs * delays[c] / M;
        s.wq[(size_t) f * C + c] = std::polar(1.0, val) / (double) C;
        s.wq[(size_t) (M - 1 - f) * C + c] = std::polar(1.0, -val) / (double) C;
      }
  } else {                                                  // :557-581
    for (int c = 0; c < C; c++) s.wq[c] = std::polar(1.0, 0.0) / (double) C;
    for (int f = 1; f < M2; f++)
      for (int c = 0; c < C; c++) {
        const double val = -2.0 * M_PI * f * delays[c] * fs / M;
        s.wq[(size_t) f * C + c] = std::polar(1.0, val) / (double) C;
        s.wq[(size_t) (M - f) * C + c] = std::polar(1.0, -val) / (double) C;
      }
    for (int c = 0; c < C; c++) { const double val = -M_PI * fs * delays[c]; s.wq[(size_t) M2 * C + c] = std::polar(1.0, val) / (double) C; }
  }
  s.haveWq = true; s.dirty = true;
}

static void blocking_matrix(const zc* d, int C, zc* B)       // NC = 1, beamformer.cc:398-479
{
  const int bs = C - 1;
  std::vector<zc> P((size_t) C * C), vec(C);
  double nrm = 0; for (int i = 0; i < C; i++) nrm += std::norm(d[i]);
  nrm = std::sqrt(nrm); nrm = nrm * nrm;
  for (int i = 0; i < C; i++) for (int j = 0; j < C; j++) P[(size_t) i * C + j] = (i == j ? 1.0 : 0.0) + (-1.0 / nrm) * std::conj(d[i]) * d[j];
  for (int k = 0; k < C * bs; k++) B[k] = zc(0, 0);
  for (int id = 0; id < bs; id++) {
    for (int i = 0; i < C; i++) vec[i] = P[(size_t) i * C + id];
    for (int jd = 0; jd < id; jd++) {
      zc ip(0, 0); for (int i = 0; i < C; i++) ip += std::conj(B[(size_t) i * bs + jd]) * vec[i];
      ip = -ip; for (int i = 0; i < C; i++) vec[i] += ip * B[(size_t) i * bs + jd];
    }
    double nv = 0; for (int i = 0; i < C; i++) nv += std::norm(vec[i]); nv = std::sqrt(nv);
    for (int i = 0; i < C; i++) B[(size_t) i * bs + id] = vec[i] * (1.0 / nv);
  }
}

static void update_wl(BfState& s, int f)          // calcSidelobeCancellerP_f / U_f: wl = B wa (beamformer.cc:761-799)
{
  const int C = s.C, bs = C - 1;
  if (s.wl.size() != (size_t) s.M * C) s.wl.assign((size_t) s.M * C, zc(0, 0));
  for (int i = 0; i < C; i++) { zc a(0, 0); for (int j = 0; j < bs; j++) a += s.B[((size_t) f * C + i) * bs + j] * s.wa[(size_t) f * bs + j]; s.wl[(size_t) f * C + i] = a; }
}

static void refresh_effective(BfState& s)
{
  const int M = s.M, C = s.C;
  // halfBandShift == true: every one of the M bins is computed on its own -- no mirror, no special bin 0 (SubbandDS::next beamformer.cc:1159-1175,
  // SubbandGSC::next :1321-1330); the snapshots then carry all M bins, [U][C][Tmax][M].  SubbandMVDR refuses the flag at construction (:2324-2327).
  const int F = s.halfBandShift ? M : M / 2 + 1;
  if (s.halfBandShift && (s.mode == 1 || s.mode == 4)) throw Error(DSR_E_ALLOCATION, "halfBandShift==true is not yet supported");
  s.eff.assign((size_t) F * C, zc(0, 0));
  if (s.mode == 0) {
    if (!s.haveWq) throw Error(DSR_E_ERROR, "call calcArrayManifoldVectorsX() once");          // beamformer.cc:1140-1143
    for (int f = 0; f < F; f++) for (int c = 0; c < C; c++) s.eff[(size_t) f * C + c] = s.wq[(size_t) f * C + c];
  } else if (s.mode == 1) {
    if (!s.haveMvdr) throw Error(DSR_E_ERROR, "call calcMVDRWeights() once");                    // :2591-2594
    s.eff = s.mvdr;
  } else if (s.mode == 4) {                                                                        // SubbandMVDRGSC::next (beamformer.cc:2764-2817)
    if (!s.haveB) throw Error(DSR_E_ERROR, "call calcArrayManifoldVectorsX() once");
    if (!s.haveMvdr) throw Error(DSR_E_ERROR, "call calcMVDRWeights() once");
    for (int c = 0; c < C; c++) s.eff[c] = s.mvdr[c];
    for (int f = 1; f < F; f++) for (int i = 0; i < C; i++) s.eff[(size_t) f * C + i] = s.mvdr[(size_t) f * C + i] - s.wl[(size_t) f * C + i];
  } else {
    if (!s.haveWq || !s.haveB) throw Error(DSR_E_ERROR, "call calcGSCWeightsX() once");          // :1310-1313
    const int bs = C - 1;
    if (!s.halfBandShift) for (int c = 0; c < C; c++) s.eff[c] = s.wq[c];                       // bin 0: wq only (:1335-1338)
    for (int f = s.halfBandShift ? 0 : 1; f < F; f++) {
      double nrm = 0;
      for (int i = 0; i < C; i++) {
        zc wl(0, 0); for (int j = 0; j < bs; j++) wl += s.B[((size_t) f * C + i) * bs + j] * s.wa[(size_t) f * bs + j];
        const zc w = s.wq[(size_t) f * C + i] - wl; s.eff[(size_t) f * C + i] = w; nrm += std::norm(w);
      }
      if (s.mode == 3) { nrm = std::sqrt(nrm); for (int i = 0; i < C; i++) s.eff[(size_t) f * C + i] /= (nrm * C); }   // :1272-1281
    }
  }
  std::vector<float2> w((size_t) F * C);
  for (size_t i = 0; i < w.size(); i++) w[i] = make_float2((float) s.eff[i].real(), (float) s.eff[i].imag());
  s.d_w.upload(w);
  if (!s.halfBandShift) {
    std::vector<float2> wT((size_t) C * (F + 1), make_float2(0.f, 0.f));
    for (int f = 0; f < F; f++) for (int c = 0; c < C; c++) wT[(size_t) c * (F + 1) + f] = w[(size_t) f * C + c];
    s.d_wT.upload(wT);
  }
  s.dirty = false;
}

// Y[u][t][f] = sum_c conj(w[f][c]) X[u][c][t][f]
__global__ __launch_bounds__(256) void k_bf_apply(const float2* __restrict__ X, const float2* __restrict__ W,
                                                  float2* __restrict__ Y, int C, int Tmax, int F, long perUtt /*Tmax*F*/)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* w = reinterpret_cast<float2*>(smem);
  for (int i = threadIdx.x; i < F * C; i += blockDim.x) w[i] = W[i];
  __syncthreads();
  const int u = blockIdx.y;
  const float2* Xu = X + (long) u * C * perUtt;
  float2* Yu = Y + (long) u * perUtt;
  for (long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; idx < perUtt; idx += (long) gridDim.x * blockDim.x) {
    const int f = (int) (idx % F);
    float2 acc = make_float2(0.f, 0.f);
    for (int c = 0; c < C; c++) {
      const float2 x = Xu[(long) c * perUtt + idx];
      const float2 wc = w[f * C + c];
      acc.x += wc.x * x.x + wc.y * x.y;       // conj(w) * x
      acc.y += wc.x * x.y - wc.y * x.x;
    }
    Yu[idx] = acc;
  }
}

}  // namespace dsr

using namespace dsr;

// SubbandGSCRLS::next + _updateActiveWeightVector2 (beamformer.cc:1554-1698).  One thread owns one (utterance, bin) and walks the frames: the
// frame's output with the weights as they stand, then Z = B^H X, the gain vector, the precision matrix and the active weights (quadratic
// constraint optional), all fp64 in the reference's order of operations; precision matrix and active weights live in a state array
// [entry][utterance x bin] (coalesced across the threads of a wave).  Every utterance starts from P0 and wa = 0.
__device__ __forceinline__ double2 cdiv_gsl2(double ar, double ai, double br, double bi)
{ const double s = 1.0 / hypot(br, bi); const double sbr = s * br, sbi = s * bi; return make_double2((ar * sbr + ai * sbi) * s, (ai * sbr - ar * sbi) * s); }
__device__ __forceinline__ double2 cmul2(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cmulc2(double2 a, double2 b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }     // a conj(b)

// (CT: compile-time channel count -- the small vectors then live in registers and the loops unroll; 0: run-time count, arrays in scratch)
// nframesArr (optional): utterance u is adapted over its first nframesArr[u] frames only and the rest of its rows is zero filled -- the reference
// stops at the stream's last frame (beamformer.cc:1552-1612); adapting on zero-padded frames would still scale P by 1/myu and wa by (I - sigma2 P).
// carry (optional, [entry][U x F]): start from / leave the adaptation state there instead of P0 and wa = 0 (carryIn) / nowhere (carryOut).
// CA: capacity of the per-thread vectors for the run-time channel count (16, or 64 for the large arrays of BASELINE configs[4]; those live in scratch).
template <int CT, bool REG, int CAP>
__global__ __launch_bounds__(64) void k_gsc_rls(const float2* __restrict__ X, const double2* __restrict__ wq, const double2* __restrict__ B,
                                                const double2* __restrict__ P0, const double* __restrict__ diagW, double2* __restrict__ state,
                                                float2* __restrict__ Y, double2* __restrict__ waOut, int U, int Crt, int Tmax, int F, double rmu,
                                                double alpha, int qctype, int adapt, int normalize, int ldsState,
                                                const int* __restrict__ nframesArr, double2* __restrict__ carry, int carryIn, int carryOut)
{
  const int C = CT ? CT : Crt;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const long tix = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= (long) U * F) return;
  const int u = (int) (tix / F), f = (int) (tix - (long) u * F), n = C - 1;
  // the adaptation state of a thread, entry e at P[e * S]: in LDS ([entry][lane], conflict free) when (n^2 + n) x 16 B x 64 lanes fit, else in memory
  const long S = ldsState ? 64 : (long) U * F;
  const float2* Xu = X + (long) u * C * Tmax * F; float2* Yu = Y + (long) u * Tmax * F;
  const double2* wqf = wq + (long) f * C; const double2* Bf = B + (long) f * C * n;
  double2* P = ldsState ? reinterpret_cast<double2*>(smem) + threadIdx.x : state + tix; double2* wa = P + (long) n * n * S;
  constexpr int CA = CT ? CT : CAP;
  double2 x[CA], w[CA], Z[CA], PH[CA], g[CA], wn[CA];
  // REG (needs CT): precision matrix and active weights in registers -- every index below is a compile-time constant after unrolling
  double2 Pr[REG ? CA : 1][REG ? CA : 1], war[REG ? CA : 1];
#define PX(i, j) (REG ? Pr[REG ? (i) : 0][REG ? (j) : 0] : P[(long) ((i) * n + (j)) * S])
#define WA(j) (REG ? war[REG ? (j) : 0] : wa[(long) (j) * S])
  const long SC = (long) U * F;                                   // the carried state's entry pitch
  if (f > 0) {
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) PX(i, j) = carryIn ? carry[(long) (i * n + j) * SC + tix] : P0[(long) f * n * n + i * n + j];
      WA(i) = carryIn ? carry[(long) (n * n + i) * SC + tix] : make_double2(0.0, 0.0);
    }
  }
  const double dw = diagW[f];
  const int Tu = nframesArr ? (nframesArr[u] < Tmax ? nframesArr[u] : Tmax) : Tmax;
  for (int t = Tu; t < Tmax; t++) Yu[(long) t * F + f] = make_float2(0.f, 0.f);
  for (int t = 0; t < Tu; t++) {
    for (int c = 0; c < C; c++) { const float2 v = Xu[((long) c * Tmax + t) * F + f]; x[c] = make_double2((double) v.x, (double) v.y); }
    double2 y = make_double2(0.0, 0.0);
    if (f == 0) { for (int c = 0; c < C; c++) { const double2 q = cmulc2(x[c], wqf[c]); y.x += q.x; y.y += q.y; } }
    else {
      double nrm = 0.0;
      for (int i = 0; i < C; i++) {
        double2 wl = make_double2(0.0, 0.0);
        for (int j = 0; j < n; j++) { const double2 q = cmul2(Bf[i * n + j], WA(j)); wl.x += q.x; wl.y += q.y; }
        w[i] = make_double2(wqf[i].x - wl.x, wqf[i].y - wl.y); nrm += w[i].x * w[i].x + w[i].y * w[i].y;
      }
      if (normalize) { nrm = sqrt(nrm) * (double) C; for (int i = 0; i < C; i++) { w[i].x /= nrm; w[i].y /= nrm; } }
      for (int c = 0; c < C; c++) { const double2 q = cmulc2(x[c], w[c]); y.x += q.x; y.y += q.y; }
    }
    Yu[(long) t * F + f] = make_float2((float) y.x, (float) y.y);
    if (f == 0 || !adapt) continue;
    for (int j = 0; j < n; j++) { double2 a = make_double2(0.0, 0.0); for (int c = 0; c < C; c++) { const double2 q = cmulc2(x[c], Bf[c * n + j]); a.x += q.x; a.y += q.y; } Z[j] = a; }
    for (int j = 0; j < n; j++) { double2 a = make_double2(0.0, 0.0); for (int i = 0; i < n; i++) { const double2 q = cmulc2(Z[i], PX(i, j)); a.x += q.x; a.y += q.y; } PH[j] = a; }
    for (int i = 0; i < n; i++) { double2 a = make_double2(0.0, 0.0); for (int j = 0; j < n; j++) { const double2 q = cmul2(PX(i, j), Z[j]); a.x += q.x; a.y += q.y; } g[i] = make_double2(a.x * rmu, a.y * rmu); }
    double2 de = make_double2(0.0, 0.0);
    for (int j = 0; j < n; j++) { const double2 q = cmulc2(Z[j], PH[j]); de.x += q.x; de.y += q.y; }
    de = make_double2(de.x * rmu + 1.0, de.y * rmu);
    for (int i = 0; i < n; i++) g[i] = cdiv_gsl2(g[i].x, g[i].y, de.x, de.y);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        const double2 o = PX(i, j), q = cmulc2(g[i], PH[j]);
        PX(i, j) = make_double2((o.x - q.x) * rmu, (o.y - q.y) * rmu);
      }
    const double2 epA = make_double2(y.x, -y.y);
    for (int i = 0; i < n; i++) {
      double2 a = make_double2(0.0, 0.0);
      for (int j = 0; j < n; j++) {
        const double2 p = PX(i, j); double2 m1 = make_double2(p.x * (-dw), p.y * (-dw)); if (i == j) m1.x += 1.0;
        const double2 q = cmul2(m1, WA(j)); a.x += q.x; a.y += q.y;
      }
      const double2 q = cmul2(g[i], epA); wn[i] = make_double2(a.x + q.x, a.y + q.y);
    }
    if (qctype == 1 || qctype == 2) {
      double nr = 0.0; for (int i = 0; i < n; i++) nr += wn[i].x * wn[i].x + wn[i].y * wn[i].y;
      nr = sqrt(nr);
      if (qctype == 1 || nr * nr >= alpha) { const double sc = alpha / nr; for (int i = 0; i < n; i++) { wn[i].x *= sc; wn[i].y *= sc; } }
    }
    for (int i = 0; i < n; i++) WA(i) = wn[i];
  }
  if (carryOut && f > 0)
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) carry[(long) (i * n + j) * SC + tix] = PX(i, j);
      carry[(long) (n * n + i) * SC + tix] = WA(i);
    }
  if (waOut && f > 0) for (int j = 0; j < n; j++) waOut[((long) u * F + f) * n + j] = WA(j);
  if (waOut && f == 0) for (int j = 0; j < n; j++) waOut[((long) u * F) * n + j] = make_double2(0.0, 0.0);
}
#undef PX
#undef WA

struct dsr_bf : BfState {};


static void gsc_rls_apply(BfState& s, const float* X, const int32_t* nframes, int U, int Tmax, float* Y, double* waOut, hipStream_t st)
{
  if (!s.haveB) throw Error(DSR_E_ERROR, "call calcGSCWeightsX() once");                                                    // beamformer.cc:1562-1565
  if (!s.haveP0) throw Error(DSR_E_ERROR, "set the precision matrix with initPrecisionMatrix() or setPrecisionMatrix()");   // :1566-1569
  if (s.halfBandShift) throw Error(DSR_E_ERROR, "not yet implemented");                                                     // :1580-1583
  if (s.C > 64) throw Error(DSR_E_DIMENSION, "SubbandGSCRLS: at most 64 channels (%d)", s.C);
  if (U <= 0 || Tmax <= 0) return;
  const int C = s.C, n = C - 1, F = s.M / 2 + 1;
  if (s.rlsDirty || s.dirty) {
    std::vector<double2> a((size_t) F * C), b((size_t) F * C * n), p((size_t) F * n * n);
    for (size_t i = 0; i < a.size(); i++) a[i] = make_double2(s.wq[i].real(), s.wq[i].imag());
    for (size_t i = 0; i < b.size(); i++) b[i] = make_double2(s.B[i].real(), s.B[i].imag());
    for (size_t i = 0; i < p.size(); i++) p[i] = make_double2(s.rlsP0[i].real(), s.rlsP0[i].imag());
    s.d_wq.upload(a); s.d_B.upload(b); s.d_P0.upload(p); s.d_diag.upload(s.rlsDiag); s.rlsDirty = false;
  }
  const long S = (long) U * F;
  const size_t ldsB = (size_t) (n * n + n) * 16 * 64; const int ldsState = (ldsB <= 150 * 1024 && !getenv("DSR_RLS_MEMSTATE")) ? 1 : 0;
  const bool regs = getenv("DSR_RLS_NOREGS") == nullptr && (C == 8 || C == 6 || C == 4);     // precision matrix + active weights in registers (C = 8: 256 VGPRs, no scratch)
  // carried state: its own array (the register / LDS variants copy in and out; the memory variant works in it directly)
  int carryIn = 0, carryOut = 0; double2* carry = nullptr;
  if (s.rlsCarry) {
    if (s.rlsHaveState && s.rlsStateU != U) throw Error(DSR_E_CONSISTENCY, "SubbandGSCRLS: the carried state holds %d streams, this call has %d (reset the state first)", s.rlsStateU, U);
    s.d_carry.reserve((size_t) S * (n * n + n));
    carry = s.d_carry.p; carryIn = s.rlsHaveState ? 1 : 0; carryOut = 1;
  }
  const bool memInPlace = !regs && !ldsState;                   // state array == working array
  if (memInPlace && !s.rlsCarry) s.d_state.reserve((size_t) S * (n * n + n));
  double2* work = memInPlace ? (s.rlsCarry ? s.d_carry.p : s.d_state.p) : (s.d_state.reserve(16), s.d_state.p);
  // in-place memory variant with carry: the working array IS the carried one: reading "carry" at start and writing it at the end are no-ops on the
  // same addresses (same [entry][U x F] layout), so the flags are passed as they are.
#define RLS_ARGS (const float2*) X, s.d_wq.p, s.d_B.p, s.d_P0.p, s.d_diag.p, work, (float2*) Y, (double2*) waOut, U, C, Tmax, F, 1.0 / s.rlsMyu, s.rlsAlpha, s.rlsQc, \
                 s.rlsAdapt ? 1 : 0, s.mode == 3 ? 1 : 0
#define RLS_TAIL nframes, carry, carryIn, carryOut
#define RLS_LAUNCH(CTV, CAPV) { if (ldsState) DSR_HIP(hipFuncSetAttribute((const void*) k_gsc_rls<CTV, false, CAPV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsB)); \
  hipLaunchKernelGGL((k_gsc_rls<CTV, false, CAPV>), dim3((unsigned) ((S + 63) / 64)), dim3(64), ldsState ? ldsB : 0, st, RLS_ARGS, ldsState, RLS_TAIL); }
#define RLS_LAUNCH_REG(CTV) hipLaunchKernelGGL((k_gsc_rls<CTV, true, 16>), dim3((unsigned) ((S + 63) / 64)), dim3(64), 0, st, RLS_ARGS, 0, RLS_TAIL);
  if (regs && C == 8) { RLS_LAUNCH_REG(8) } else if (regs && C == 6) { RLS_LAUNCH_REG(6) } else if (regs && C == 4) { RLS_LAUNCH_REG(4) }
  else if (C == 8) RLS_LAUNCH(8, 16) else if (C == 4) RLS_LAUNCH(4, 16) else if (C == 6) RLS_LAUNCH(6, 16) else if (C <= 16) RLS_LAUNCH(0, 16) else RLS_LAUNCH(0, 64)
#undef RLS_LAUNCH_REG
#undef RLS_LAUNCH
#undef RLS_ARGS
#undef RLS_TAIL
  DSR_HIP(hipGetLastError());
  if (s.rlsCarry) { s.rlsHaveState = true; s.rlsStateU = U; }
}

extern "C" {

dsr_status dsr_bf_create(int fftLen, int chanN, int halfBandShift, dsr_bf** out)
{
  return guard([&] {
    if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (fftLen < 2 || (fftLen & 1) || chanN < 1) throw Error(DSR_E_DIMENSION, "bad fftLen=%d chanN=%d", fftLen, chanN);
    dsr_bf* s = new dsr_bf(); s->M = fftLen; s->C = chanN; s->halfBandShift = halfBandShift; *out = s;
  });
}
void dsr_bf_destroy(dsr_bf* s) { delete s; }
int dsr_bf_fft_len(const dsr_bf* s) { return s->M; }
int dsr_bf_chan_n(const dsr_bf* s) { return s->C; }
int dsr_bf_half_band_shift(const dsr_bf* s) { return s->halfBandShift ? 1 : 0; }
int dsr_bf_is_adaptive(const dsr_bf* s) { return (s && s->rlsOn) ? 1 : 0; }
int dsr_bf_bins(const dsr_bf* s) { return s->halfBandShift ? s->M : s->M / 2 + 1; }

dsr_status dsr_bf_calc_array_manifold(dsr_bf* s, double fs, const double* delays)
{ return guard([&] { if (!s || !delays) throw Error(DSR_E_PARAMETER, "null argument"); calc_mainlobe(*s, fs, delays); }); }

dsr_status dsr_calc_delays_polar2(float azimuth, float elevation, const double* micPos, int C, double* delays)
{
  return guard([&] {
    if (!micPos || !delays) throw Error(DSR_E_PARAMETER, "null argument");
    // superdirectiveBeamformer.cc:118-137 -- float arithmetic, float sin/cos overloads, mm/s
    const float c_x = -sinf(elevation) * cosf(azimuth), c_y = -sinf(elevation) * sinf(azimuth), c_z = -cosf(elevation);
    for (int i = 0; i < C; i++) {
      const float x = (float) micPos[3 * i], y = (float) micPos[3 * i + 1], z = (float) micPos[3 * i + 2];
      const float t = (c_x * x + c_y * y + c_z * z) / 343740.0;
      delays[i] = t;
    }
  });
}

dsr_status dsr_bf_set_diffuse_noise_model(dsr_bf* s, const double* mp, double fs, double sspeed)
{
  return guard([&] {
    if (!s || !mp) throw Error(DSR_E_PARAMETER, "null argument");
    const int C = s->C, M = s->M, F = M / 2 + 1;
    std::vector<double> dm((size_t) C * C, 0.0);
    for (int m = 0; m < C; m++) for (int n = 0; n < m; n++) {
      const double dx = mp[3*m] - mp[3*n], dy = mp[3*m+1] - mp[3*n+1], dz = mp[3*m+2] - mp[3*n+2];
      dm[(size_t) m * C + n] = std::sqrt(dx * dx + dy * dy + dz * dz);
    }
    s->R.assign((size_t) F * C * C, zc(0, 0));
    for (int f = 0; f < F; f++) {
      const double odc = 2.0 * fs * f / (M * sspeed);
      zc* Rf = &s->R[(size_t) f * C * C];
      for (int m = 0; m < C; m++) for (int n = 0; n < m; n++) Rf[(size_t) m * C + n] = zc(sinc_pi(odc * dm[(size_t) m * C + n]), 0.0);
      for (int m = 0; m < C; m++) Rf[(size_t) m * C + m] = zc(1.0, 0.0);
      for (int m = 0; m < C; m++) for (int n = m + 1; n < C; n++) Rf[(size_t) m * C + n] = Rf[(size_t) n * C + m];
    }
    s->haveR = true;
  });
}
dsr_status dsr_bf_divide_nondiagonal(dsr_bf* s, float myu)
{
  return guard([&] {
    if (!s || !s->haveR) throw Error(DSR_E_ERROR, "Construct first a noise covariance matrix");
    const int C = s->C, F = s->M / 2 + 1;
    for (int f = 0; f < F; f++) for (int x = 0; x < C; x++) for (int y = 0; y < C; y++)
      if (x != y) s->R[((size_t) f * C + x) * C + y] /= zc(1.0 + myu, 0.0);
  });
}
dsr_status dsr_bf_diagonal_loading(dsr_bf* s, float w)
{
  return guard([&] {
    if (!s || !s->haveR) throw Error(DSR_E_ERROR, "Construct first a noise covariance matrix");
    const int C = s->C, F = s->M / 2 + 1;
    for (int f = 0; f < F; f++) for (int c = 0; c < C; c++) s->R[((size_t) f * C + c) * C + c] += (double) w;
  });
}
dsr_status dsr_bf_set_noise_matrix(dsr_bf* s, int f, const double* Rnn)
{
  return guard([&] {
    if (!s || !Rnn) throw Error(DSR_E_PARAMETER, "null argument");
    const int C = s->C, F = s->M / 2 + 1;
    if (f < 0 || f >= F) throw Error(DSR_E_INDEX, "frequency bin %d out of range", f);
    if (!s->haveR) { s->R.assign((size_t) F * C * C, zc(0, 0)); s->haveR = true; }
    for (int i = 0; i < C * C; i++) s->R[(size_t) f * C * C + i] = zc(Rnn[2 * i], Rnn[2 * i + 1]);
  });
}
// pseudoinverse(A, invA, dThreshold) (beamformer.cc:253-305) by itself: host-side, no device needed
dsr_status dsr_pseudoinverse(const double* A, int rows, int cols, float dThreshold, double* invA, int* ok, float* svals)
{
  return guard([&] {
    if (!A || !invA || rows < 1 || cols < 1) throw Error(DSR_E_PARAMETER, "bad argument");
    const bool r = linpack::pseudoinverse(reinterpret_cast<const zc*>(A), reinterpret_cast<zc*>(invA), rows, cols, dThreshold, svals);
    if (ok) *ok = r ? 1 : 0;
  });
}
dsr_status dsr_bf_calc_mvdr_weights(dsr_bf* s, double fs, double thr)
{
  (void) fs;
  return guard([&] {
    if (!s) throw Error(DSR_E_PARAMETER, "null argument");
    if (!s->haveR) throw Error(DSR_E_ALLOCATION, "Set a spatial spectral matrix before calling calcMVDRWeights()");
    if (!s->haveWq) throw Error(DSR_E_ERROR, "call calcArrayManifoldVectorsX() once");
    const int C = s->C, F = s->M / 2 + 1;
    s->mvdr.assign((size_t) F * C, zc(0, 0));
    std::vector<zc> invR((size_t) C * C), tmpH(C);
    for (int c = 0; c < C; c++) s->mvdr[c] = zc(1.0, 0.0);                          // :2413-2415
    for (int f = 1; f < F; f++) {
      const zc* d = &s->wq[(size_t) f * C];
      if (!pseudoinverse_cf(&s->R[(size_t) f * C * C], invR.data(), C, (float) thr)) {
        for (int i = 0; i < C * C; i++) invR[i] = zc(0, 0);
        for (int c = 0; c < C; c++) invR[(size_t) c * C + c] = zc(1, 0);
      }
      for (int i = 0; i < C; i++) { zc acc(0, 0); for (int j = 0; j < C; j++) acc += std::conj(invR[(size_t) j * C + i]) * d[j]; tmpH[i] = acc; }
      zc Lambda(0, 0); for (int i = 0; i < C; i++) Lambda += std::conj(tmpH[i]) * d[i];
      const zc norm = Lambda * (double) C;
      // gsl_complex_div (GSL complex/math.c): scale by 1/|b| first -- restated so that the last bit does not depend on a runtime's __divdc3
      const double sN = 1.0 / std::hypot(norm.real(), norm.imag()), sbr = sN * norm.real(), sbi = sN * norm.imag();
      for (int c = 0; c < C; c++) s->mvdr[(size_t) f * C + c] = zc((tmpH[c].real() * sbr + tmpH[c].imag() * sbi) * sN, (tmpH[c].imag() * sbr - tmpH[c].real() * sbi) * sN);
    }
    s->haveMvdr = true; s->dirty = true;
  });
}
dsr_status dsr_bf_calc_gsc_weights(dsr_bf* s, double fs, const double* delays)
{
  return guard([&] {
    if (!s || !delays) throw Error(DSR_E_PARAMETER, "null argument");
    if (s->C <= 1) throw Error(DSR_E_DIMENSION, "The number of channels must be > 1 but it is %d", s->C);
    calc_mainlobe(*s, fs, delays);
    const int C = s->C, M = s->M, bs = C - 1;
    s->B.assign((size_t) M * C * bs, zc(0, 0)); s->wa.assign((size_t) M * bs, zc(0, 0)); s->wl.assign((size_t) M * C, zc(0, 0));
    for (int f = 0; f < M; f++) blocking_matrix(&s->wq[(size_t) f * C], C, &s->B[(size_t) f * C * bs]);
    s->haveB = true; s->dirty = true;
  });
}
dsr_status dsr_bf_set_active_weights(dsr_bf* s, int f, const double* packed)
{
  return guard([&] {
    if (!s || !packed) throw Error(DSR_E_PARAMETER, "null argument");
    if (!s->haveB) throw Error(DSR_E_ERROR, "call calcGSCWeightsX() once");
    if (f < 0 || f >= s->M) throw Error(DSR_E_DIMENSION, "Must be a frequency bin %d < the length of FFT %d", f, s->M);
    for (int c = 0; c < s->C - 1; c++) s->wa[(size_t) f * (s->C - 1) + c] = zc(packed[2 * c], packed[2 * c + 1]);
    update_wl(*s, f);
    s->dirty = true;
  });
}
dsr_status dsr_bf_zero_active_weights(dsr_bf* s)
{
  return guard([&] {
    if (!s || !s->haveB) throw Error(DSR_E_ERROR, "call calcGSCWeightsX() once");
    std::fill(s->wa.begin(), s->wa.end(), zc(0, 0)); for (int f = 0; f < s->M; f++) update_wl(*s, f); s->dirty = true;
  });
}
dsr_status dsr_bf_select(dsr_bf* s, int mode)
{ return guard([&] { if (!s || mode < 0 || mode > 4) throw Error(DSR_E_PARAMETER, "bad mode"); s->mode = mode; s->dirty = true; }); }

dsr_status dsr_bf_get(const dsr_bf* cs, int kind, double* out, size_t nd)
{
  return guard([&] {
    dsr_bf* s = const_cast<dsr_bf*>(cs);
    if (!s || !out) throw Error(DSR_E_PARAMETER, "null argument");
    const std::vector<zc>* v = nullptr;
    if (kind == 4) { if (s->dirty) { require_device(); refresh_effective(*s); } v = &s->eff; }
    else v = kind == 0 ? &s->wq : kind == 1 ? &s->mvdr : kind == 2 ? &s->R : kind == 3 ? &s->B : nullptr;
    if (!v) throw Error(DSR_E_PARAMETER, "bad kind %d", kind);
    if (nd < 2 * v->size()) throw Error(DSR_E_DIMENSION, "output holds %zu doubles, need %zu", nd, 2 * v->size());
    for (size_t i = 0; i < v->size(); i++) { out[2 * i] = (*v)[i].real(); out[2 * i + 1] = (*v)[i].imag(); }
  });
}

// the weights k_bf_apply would use, channel-major [C][M/2+2] on the device, for the fused analysis + beamformer kernel (k_filterbank.hip); null when the output is
// not a fixed linear combination of the channels (SubbandGSCRLS adapting) or when all M bins are computed (halfBandShift)
namespace dsr { const float2* bf_fixed_weights_dev(dsr_bf* s) { if (!s || s->rlsOn || s->halfBandShift) return nullptr; if (s->dirty) refresh_effective(*s); return s->d_wT.p; } }

dsr_status dsr_bf_apply(dsr_bf* s, const float* X, int U, int Tmax, float* Y, void* stream)
{
  return guard([&] {
    if (!s || !X || !Y) throw Error(DSR_E_PARAMETER, "null argument");
    require_device();
    if (s->rlsOn) { gsc_rls_apply(*s, X, nullptr, U, Tmax, Y, nullptr, (hipStream_t) stream); return; }
    if (s->dirty) refresh_effective(*s);
    if (U <= 0 || Tmax <= 0) return;
    const int F = s->halfBandShift ? s->M : s->M / 2 + 1; const long perUtt = (long) Tmax * F;
    const size_t lds = sizeof(float2) * (size_t) F * s->C;
    if (lds > 160 * 1024) throw Error(DSR_E_DIMENSION, "beamformer weights need %zu bytes of LDS", lds);
    DSR_HIP(hipFuncSetAttribute((const void*) k_bf_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    int gx = cdiv(perUtt, 256 * 4); if (gx < 1) gx = 1; if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_bf_apply, dim3(gx, U), dim3(256), lds, (hipStream_t) stream, (const float2*) X, s->d_w.p, (float2*) Y,
                       s->C, Tmax, F, perUtt);
    DSR_HIP(hipGetLastError());
  });
}


// SubbandGSCRLS(fftLen, halfBandShift, myu, sigma2) (beamformer.h:230-262).  rls_config switches dsr_bf_apply to the recursive-least-squares
// GSC (the object must hold GSC weights: calcGSCWeights); sigma2 is the constructor's diagonal weight of the active-weight update.
dsr_status dsr_bf_rls_config(dsr_bf* s, float myu, float sigma2)
{
  return guard([&] {
    if (!s) throw Error(DSR_E_PARAMETER, "null argument");
    if (!(myu > 0.0f)) throw Error(DSR_E_PARAMETER, "the forgetting factor must be positive (%g)", (double) myu);
    s->rlsOn = true; s->rlsMyu = (double) myu; s->rlsDiag.assign((size_t) s->M / 2 + 1, (double) sigma2); s->rlsDirty = true;
    if (s->mode < 2) s->mode = 2;
  });
}
// initPrecisionMatrix(sigma2): P = I / sigma2 for every bin, active weights zeroed (beamformer.cc:1526-1538)
dsr_status dsr_bf_rls_init_precision(dsr_bf* s, float sigma2)
{
  return guard([&] {
    if (!s) throw Error(DSR_E_PARAMETER, "null argument");
    if (!s->haveB) throw Error(DSR_E_ERROR, "call calcGSCWeightsX() once");
    const int n = s->C - 1, F = s->M / 2 + 1;
    s->rlsP0.assign((size_t) F * n * n, zc(0, 0));
    for (int f = 0; f < F; f++) for (int i = 0; i < n; i++) s->rlsP0[((size_t) f * n + i) * n + i] = zc((double) (1 / sigma2), 0.0);
    std::fill(s->wa.begin(), s->wa.end(), zc(0, 0));
    s->haveP0 = true; s->rlsDirty = true; s->dirty = true; s->rlsHaveState = false;
  });
}
// setPrecisionMatrix(fbinX, Pz) (:1540-1552); Pz [C-1][C-1] complex128
dsr_status dsr_bf_rls_set_precision(dsr_bf* s, int fbinX, const double* Pz)
{
  return guard([&] {
    if (!s || !Pz) throw Error(DSR_E_PARAMETER, "null argument");
    if (!s->haveB) throw Error(DSR_E_ERROR, "call calcGSCWeightsX() once");
    const int n = s->C - 1, F = s->M / 2 + 1;
    if (fbinX < 0 || fbinX >= F) throw Error(DSR_E_DIMENSION, "Must be a frequency bin %d <= %d", fbinX, F - 1);
    if (s->rlsP0.size() != (size_t) F * n * n) s->rlsP0.assign((size_t) F * n * n, zc(0, 0));
    for (int e = 0; e < n * n; e++) s->rlsP0[(size_t) fbinX * n * n + e] = zc(Pz[2 * e], Pz[2 * e + 1]);
    s->haveP0 = true; s->rlsDirty = true; s->rlsHaveState = false;
  });
}
dsr_status dsr_bf_rls_quadratic_constraint(dsr_bf* s, float alpha, int qctype)
{ return guard([&] { if (!s || qctype < 0 || qctype > 2) throw Error(DSR_E_PARAMETER, "bad quadratic constraint type"); s->rlsAlpha = (double) alpha; s->rlsQc = qctype; }); }
dsr_status dsr_bf_rls_adapt(dsr_bf* s, int flag) { return guard([&] { if (!s) throw Error(DSR_E_PARAMETER, "null argument"); s->rlsAdapt = flag != 0; }); }
// batch entry with the final active weights: wa_out_dev (optional) [U][M/2+1][C-1] complex128
dsr_status dsr_bf_gsc_rls(dsr_bf* s, const float* X, const int32_t* nframes_dev, int U, int Tmax, float* Y, double* wa_out_dev, void* stream)
{
  return guard([&] {
    if (!s || !X || !Y) throw Error(DSR_E_PARAMETER, "null argument");
    if (!s->rlsOn) throw Error(DSR_E_ERROR, "not a SubbandGSCRLS object: call dsr_bf_rls_config first");
    require_device();
    gsc_rls_apply(*s, X, nframes_dev, U, Tmax, Y, wa_out_dev, (hipStream_t) stream);
  });
}
// dsr_bf_apply with per-utterance frame counts: rows t >= nframes[u] are zero; an adapting (RLS) object stops adapting there
dsr_status dsr_bf_apply_frames(dsr_bf* s, const float* X, const int32_t* nframes_dev, int U, int Tmax, float* Y, void* stream)
{
  if (s && s->rlsOn) return dsr_bf_gsc_rls(s, X, nframes_dev, U, Tmax, Y, nullptr, stream);
  return dsr_bf_apply(s, X, U, Tmax, Y, stream);               // fixed weights: the padded rows of X are zero, so are the outputs
}
// carry on: every call continues from the precision matrices and active weights the previous call left (same U); the first call after
// rls_reset_state / rls_init_precision / rls_set_precision starts from those matrices and zero weights
dsr_status dsr_bf_rls_carry(dsr_bf* s, int on)
{ return guard([&] { if (!s) throw Error(DSR_E_PARAMETER, "null argument"); s->rlsCarry = on != 0; if (!on) s->rlsHaveState = false; }); }
dsr_status dsr_bf_rls_reset_state(dsr_bf* s)
{ return guard([&] { if (!s) throw Error(DSR_E_PARAMETER, "null argument"); s->rlsHaveState = false; }); }


// SubbandMVDRGSC (beamformer.h:394-425, beamformer.cc:2637-2817): the MVDR vector as the quiescent weight of a GSC whose active weights are
// set from outside.  calcBlockingMatrix1: a fresh weight object with the delay-and-sum vectors and their blocking matrices (:2671-2676) -- the
// same state as calcGSCWeights; calcBlockingMatrix2: fresh object, bins 1..M/2 get the MVDR vector as quiescent vector and its blocking
// matrix (:2682-2706), everything else stays zero.
dsr_status dsr_bf_calc_blocking_matrix2(dsr_bf* s)
{
  return guard([&] {
    if (!s) throw Error(DSR_E_PARAMETER, "null argument");
    if (!s->haveMvdr) throw Error(DSR_E_ERROR, "You have to call calcMVDRWeights() first");
    if (s->halfBandShift) throw Error(DSR_E_ERROR, "Not yet implemented");
    const int C = s->C, M = s->M, bs = C - 1;
    s->wq.assign((size_t) M * C, zc(0, 0)); s->B.assign((size_t) M * C * bs, zc(0, 0)); s->wa.assign((size_t) M * bs, zc(0, 0)); s->wl.assign((size_t) M * C, zc(0, 0));
    for (int f = 1; f <= M / 2; f++) {
      for (int c = 0; c < C; c++) s->wq[(size_t) f * C + c] = s->mvdr[(size_t) f * C + c];
      blocking_matrix(&s->wq[(size_t) f * C], C, &s->B[(size_t) f * C * bs]);
    }
    s->haveWq = true; s->haveB = true; s->dirty = true; s->rlsDirty = true;
  });
}
// upgradeBlockingMatrix (:2708-2727): blocking matrices orthogonal to the entire weight wq - wl, bins 1..M-1; the cached wl stays as it is
dsr_status dsr_bf_upgrade_blocking_matrix(dsr_bf* s)
{
  return guard([&] {
    if (!s || !s->haveB) throw Error(DSR_E_ERROR, "call calcGSCWeightsX() once");
    const int C = s->C, M = s->M, bs = C - 1; std::vector<zc> w(C);
    if (s->wl.size() != (size_t) M * C) s->wl.assign((size_t) M * C, zc(0, 0));
    for (int f = 1; f < M; f++) {
      for (int c = 0; c < C; c++) w[c] = s->wq[(size_t) f * C + c] - s->wl[(size_t) f * C + c];
      blocking_matrix(w.data(), C, &s->B[(size_t) f * C * bs]);
    }
    s->dirty = true; s->rlsDirty = true;
  });
}
// blockingMatrixOutput(outChanX) (:2729-2753) for a batch: Y[u][t][f] = B_f[:, outChanX]^H X[u][:][t][f], f = 0..M/2
dsr_status dsr_bf_blocking_matrix_output(dsr_bf* s, const float* X, int U, int Tmax, int outChanX, float* Y, void* stream)
{
  return guard([&] {
    if (!s || !X || !Y) throw Error(DSR_E_PARAMETER, "null argument");
    if (!s->haveB) throw Error(DSR_E_ERROR, "call calcGSCWeightsX() once");
    const int C = s->C, bs = C - 1, F = s->M / 2 + 1;
    if (outChanX < 0 || outChanX >= bs) throw Error(DSR_E_INDEX, "blocking matrix column %d of %d", outChanX, bs);
    require_device();
    if (U <= 0 || Tmax <= 0) return;
    std::vector<float2> w((size_t) F * C);
    for (int f = 0; f < F; f++) for (int c = 0; c < C; c++) { const zc b = s->B[((size_t) f * C + c) * bs + outChanX]; w[(size_t) f * C + c] = make_float2((float) b.real(), (float) b.imag()); }
    DevBuf<float2>& dW = s->d_wB.at((hipStream_t) stream); dW.upload(w, (hipStream_t) stream);
    const long perUtt = (long) Tmax * F; const size_t lds = sizeof(float2) * (size_t) F * C;
    DSR_HIP(hipFuncSetAttribute((const void*) k_bf_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    int gx = cdiv(perUtt, 256 * 4); if (gx < 1) gx = 1; if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_bf_apply, dim3(gx, U), dim3(256), lds, (hipStream_t) stream, (const float2*) X, dW.p, (float2*) Y, C, Tmax, F, perUtt);
    DSR_HIP(hipGetLastError());
  });
}

}  // extern "C"
