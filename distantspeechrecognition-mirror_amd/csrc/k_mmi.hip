// csrc/k_mmi.hip -- SubbandMMI: one generalized sidelobe canceller per sound source, the target's output post-filtered (Zelinski) and
// optionally masked against the other sources' outputs.
//
// Replaces SubbandMMI (btk/beamformer/beamformer.h:264-312, beamformer.cc:1753-2319) and what it calls for its weights:
//   calcMainlobe / calcMainlobeN      beamformer.cc:531-594, 631-753      calcNullBeamformer :314-394, putInverseMat22 :202-242
//   _calcBlockingMatrix (any NC)      :398-479                             calcSidelobeCancellerP_f/U_f :761-799, scaling :1455-1490
//   next / calcInterferenceOutputs / binaryMasking :1973-2319              ZelinskiFilter(_f) btk/postfilter/postfilter.cc:59-221
// Weight design is host set-up work as in the reference (fp64, the demixing-matrix scaling through the restated LINPACK csvdc).  The frame
// loop is k_mmi: one workgroup per utterance, a thread per frequency bin walking the frames -- the post-filter's spectral densities and the
// mask's averaged output are first-order recursions over time, independent across bins (the averaged output couples neighbouring bins only
// when useBinaryMask's fwidth > 1: those frames take a workgroup barrier and one thread does the bins in order, as the reference's loop).
// fp64 arithmetic in the reference's order on the pipe's complex64 snapshots (-ffp-contract=off).
//
// The TYPE_APAB post-filter (beamformer.cc:2047-2049, 2177-2179; ApabFilter / ApabFilter_f postfilter.cc:225-340, always called with channelX =
// chanN/2) filters the bins below fftLen/2 only (with halfBandShift it mirrors the weights onto fftLen-1-k): without halfBandShift its output vector
// is not conjugate-symmetric, so in that configuration the kernel writes all fftLen bins of a frame (dsr_mmi_out_bins) -- the upper ones as the
// reference leaves them: the conjugate of the UNFILTERED lower bin, or, where the mask struck, the mask's value itself, unconjugated (:2302-2304).
// In every other configuration the half spectrum is handed on and the mask's averaged value lands in bin k only.
#include "common.h"
#include "svd_linpack.h"
#include <complex>
#include <cmath>

namespace dsr {

typedef std::complex<double> zc;

namespace {

// GSL's complex product / quotient (complex/math.c), so that values agree with the reference to rounding
inline zc gmul(zc a, zc b) { return zc(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real()); }
inline zc gdiv(zc a, zc b)
{
  const double s = 1.0 / std::hypot(b.real(), b.imag()); const double sbr = s * b.real(), sbi = s * b.imag();
  return zc((a.real() * sbr + a.imag() * sbi) * s, (a.imag() * sbr - a.real() * sbi) * s);
}
inline zc gpolar(double r, double th) { return zc(r * std::cos(th), r * std::sin(th)); }
inline double gabs2(zc a) { return a.real() * a.real() + a.imag() * a.imag(); }

struct SrcW { std::vector<zc> wq, B, wa, wl, ta; };     // [M][C], [M][C][C-NC], [M][C-NC], [M][C], [M][C]

// _calcBlockingMatrix (beamformer.cc:398-479)
bool blocking_matrix_nc(const zc* d, int C, int NC, zc* B)
{
  const int bs = C - NC;
  if (bs <= 0) return false;
  std::vector<zc> P((size_t) C * C), vec(C);
  double nrm = 0; for (int i = 0; i < C; i++) nrm += gabs2(d[i]);
  nrm = std::sqrt(nrm); nrm = nrm * nrm;
  for (int i = 0; i < C; i++) for (int j = 0; j < C; j++) P[(size_t) i * C + j] = zc(i == j ? 1.0 : 0.0, 0.0) + gmul(gmul(zc(-1.0 / nrm, 0.0), std::conj(d[i])), d[j]);
  for (int k = 0; k < C * bs; k++) B[k] = zc(0, 0);
  for (int id = 0; id < bs; id++) {
    for (int i = 0; i < C; i++) vec[i] = P[(size_t) i * C + id];
    for (int jd = 0; jd < id; jd++) {
      zc ip(0, 0); for (int i = 0; i < C; i++) ip += gmul(std::conj(B[(size_t) i * bs + jd]), vec[i]);
      ip = zc(ip.real() * -1.0, ip.imag() * -1.0);
      for (int i = 0; i < C; i++) vec[i] += gmul(ip, B[(size_t) i * bs + jd]);
    }
    double nv = 0; for (int i = 0; i < C; i++) nv += gabs2(vec[i]);
    nv = std::sqrt(nv);
    for (int i = 0; i < C; i++) B[(size_t) i * bs + id] = zc(vec[i].real() * (1.0 / nv), vec[i].imag() * (1.0 / nv));
  }
  return true;
}

void put_inverse_mat22(zc* mat)                          // beamformer.cc:202-242
{
  const double beta = 0.01;
  zc m00 = mat[0], m11 = mat[3], m01 = mat[1], m10 = mat[2];
  zc det = gmul(m00, m11) - gmul(m01, m10);
  if (std::hypot(det.real(), det.imag()) < 1.0E-07) {
    m00 = zc(m00.real() + beta, m00.imag()); m11 = zc(m11.real() + beta, m11.imag());
    det = gmul(m00, m11) - gmul(m01, m10);
  }
  mat[0] = gdiv(m11, det); mat[3] = gdiv(m00, det);
  const zc a = gdiv(m01, det), b = gdiv(m10, det);
  mat[1] = zc(a.real() * -1.0, a.imag() * -1.0); mat[2] = zc(b.real() * -1.0, b.imag() * -1.0);
}

// calcNullBeamformer (:314-394): wt <- Cm inv(Cm^H Cm) e_0 with Cm = [wt, the interferers' manifolds]
void calc_null_beamformer(zc* wt, const std::vector<std::vector<zc>>& pWj, int C, int NC)
{
  std::vector<zc> Cm((size_t) C * NC), inv((size_t) NC * NC), v(NC);
  for (int i = 0; i < C; i++) { Cm[(size_t) i * NC] = wt[i]; for (int j = 1; j < NC; j++) Cm[(size_t) i * NC + j] = pWj[j - 1][i]; }
  for (int a = 0; a < NC; a++) for (int b = 0; b < NC; b++) { zc acc(0, 0); for (int i = 0; i < C; i++) acc += gmul(std::conj(Cm[(size_t) i * NC + a]), Cm[(size_t) i * NC + b]); inv[(size_t) a * NC + b] = acc; }
  if (NC != 2) { std::vector<zc> out((size_t) NC * NC); linpack::pseudoinverse(inv.data(), out.data(), NC, NC, 1.0E-8f); inv = out; }   // the result is taken whatever it returns
  else put_inverse_mat22(inv.data());
  for (int a = 0; a < NC; a++) { zc acc(0, 0); for (int b = 0; b < NC; b++) acc += gmul(inv[(size_t) a * NC + b], zc(b == 0 ? 1.0 : 0.0, 0.0)); v[a] = acc; }
  for (int i = 0; i < C; i++) { zc acc(0, 0); for (int a = 0; a < NC; a++) acc += gmul(Cm[(size_t) i * NC + a], v[a]); wt[i] = acc; }
}

// scaling (:1455-1490): W [Ms][N] <- diag(Wp[N/2][i]) W, Wp the pseudo-inverse.  (For Ms < N the shipped pseudoinverse loops past the
// singular values and left vectors csvdc produced, :283-297; those terms are zero here.)
void scaling(zc* W, int Ms, int N, float thr)
{
  std::vector<zc> Wp((size_t) N * Ms);
  linpack::pseudoinverse(W, Wp.data(), Ms, N, thr);
  const int stdsnsr = N / 2;
  for (int i = 0; i < Ms; i++) for (int j = 0; j < N; j++) W[(size_t) i * N + j] = gmul(Wp[(size_t) stdsnsr * Ms + i], W[(size_t) i * N + j]);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// The frame loop.  X [U][C][Tmax][F] complex64 snapshots, Y [U][Tmax][F].  weff / wup / mani: [S][F][C] fp64 complex (entire weight
// wq - wl with bin 0's wq alone when !hbs; upper branch wq; the post-filter's steering vectors).  csd: [S][U][C*C][F] spectral densities
// (only the i <= j entries are used).  taScr: [U][F][C] scratch for arrays above 16 channels.
template <int CT>
__global__ __launch_bounds__(256) void k_mmi(const float2* __restrict__ X, const int* __restrict__ nframesArr, const double2* __restrict__ weff,
                                             const double2* __restrict__ wup, const double2* __restrict__ mani, double2* __restrict__ csd,
                                             double2* __restrict__ taScr, float2* __restrict__ Y, int U, int C, int Tmax, int F, int S, int target,
                                             int hbs, int pfType, double alphaCfg, int useMask, int maskType, double avgFactor, int fwidth, int M, int Fout)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* avgOut = reinterpret_cast<double2*>(smem);             // [F] _avgOutput
  double2* outS = avgOut + F;                                     // [F] this frame's output (fwidth > 1 only)
  double2* mirS = outS + F;                                       // [F] what bin M - f of a full output frame holds unless the mask strikes (fwidth > 1 only)
  int* maskS = reinterpret_cast<int*>(mirS + F);                  // [F] target weaker than an interferer
  const int u = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int T = nframesArr[u] < Tmax ? nframesArr[u] : Tmax;
  const float2* Xu = X + (size_t) u * C * Tmax * F;
  float2* Yu = Y + (size_t) u * Tmax * Fout;
  for (int f = tid; f < F; f += nthr) avgOut[f] = make_double2(0.0, 0.0);
  __syncthreads();
  const bool apab = (pfType & 0x04) != 0;                         // tested first (beamformer.cc:2047)
  const bool zel = !apab && ((pfType & 0x01) || (pfType & 0x02));
  const bool fullOut = Fout > F;                                  // APAB without halfBandShift: all M bins of a frame are written
  const int M2 = M / 2;
  const bool wide = useMask && avgFactor >= 0.0 && fwidth > 1;
  const int NE = C * C;
  double2 taR[CT > 0 ? CT : 1];


  for (int t = 0; t < Tmax; t++) {
    if (t >= T) { for (int f = tid; f < Fout; f += nthr) Yu[(size_t) t * Fout + f] = make_float2(0.f, 0.f); continue; }   // (uniform: T is per utterance)
    const int frameX = t - 1;                                      // _frameX when next() runs (FrameResetX = -1)
    const double alpha = (frameX > 0) ? alphaCfg : 0.0;            // beamformer.cc:2042-2045
    const int type = (frameX < 0) ? 0 : pfType;                    // MINFRAMES 0 (:1136): the first frame only updates the densities
    for (int f = tid; f < F; f += nthr) {
      // ---- outputs of the target and, for the mask, of the other sources
      double2* ta = CT > 0 ? taR : taScr + ((size_t) u * F + f) * C;
      double tr = 0.0, ti = 0.0;                                   // target output
      double tgtPow = 0.0, maxPow = 0.0;
      auto dotAt = [&](const double2* w, const int fb, double& yr, double& yi) {   // sum_c conj(w_c) x_c at bin fb, products as gsl_complex_mul
        yr = 0.0; yi = 0.0;
        for (int c = 0; c < C; c++) {
          const float2 x = Xu[((size_t) c * Tmax + t) * F + fb]; const double wr = w[c].x, wi = -w[c].y;
          yr += wr * (double) x.x - wi * (double) x.y; yi += wr * (double) x.y + wi * (double) x.x;
        }
      };
      auto dot = [&](const double2* w, double& yr, double& yi) { dotAt(w, f, yr, yi); };
      // ApabFilter_f (postfilter.cc:225-264) for source s at bin fb, whose beamformed value there is (yr, yi): |y|^2 over the power of the
      // time-aligned channel chanN/2, cut at 1
      auto apabW = [&](const int s, const int fb, const double yr, const double yi) -> double {
        const int ch = C / 2;
        const double2 d = mani[((size_t) s * F + fb) * C + ch]; const float2 x = Xu[((size_t) ch * Tmax + t) * F + fb];
        const double dr = d.x, di = -d.y;
        const double pr = dr * (double) x.x - di * (double) x.y, pi = dr * (double) x.y + di * (double) x.x;
        const double pxx = pr * pr + pi * pi, pyy = yr * yr + yi * yi;
        double W = pyy / pxx;
        if (W >= 1.0) W = 1.0;
        if (W <= -1.0) W = -1.0;
        return W;
      };
      // ApabFilter (postfilter.cc:275-340) on bin f of source s's vector (computed with the weights wsel [S][F][C]): bins below M/2 take their own
      // weight; with halfBandShift bin f >= M/2 takes the weight of bin M-1-f (from that bin's snapshot, manifold and beamformed value); without it
      // bins >= M/2 are left alone
      auto apabApply = [&](const int s, const double2* wsel, double& yr, double& yi) {
        double W;
        if (f < M2) W = apabW(s, f, yr, yi);
        else if (hbs) { const int fb = M - 1 - f; double br, bi; dotAt(wsel + ((size_t) s * F + fb) * C, fb, br, bi); W = apabW(s, fb, br, bi); }
        else return;
        const double a = W * yr - 0.0 * yi, b = W * yi + 0.0 * yr; yr = a; yi = b;
      };
      auto pfw = [&](int s) -> double {                            // ZelinskiFilter for source s at this bin: returns the weight (1 when unused)
        if (!zel) return 1.0;
        const double2* d = mani + ((size_t) s * F + f) * C;
        for (int c = 0; c < C; c++) {                              // TimeAlignment: conj(d_c) x_c
          const float2 x = Xu[((size_t) c * Tmax + t) * F + f]; const double dr = d[c].x, di = -d[c].y;
          ta[c] = make_double2(dr * (double) x.x - di * (double) x.y, dr * (double) x.y + di * (double) x.x);
        }
        double2* st = csd + ((size_t) s * U + u) * NE * F + f;
        double sr = 0.0, si = 0.0;
        for (int i = 0; i < C - 1; i++)
          for (int j = i + 1; j < C; j++) {
            const double ar = ta[i].x, ai = ta[i].y, br = ta[j].x, bi = -ta[j].y;
            const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
            double er = pr, ei = pi;
            const size_t e = (size_t) (i * C + j) * F;
            if (alpha > 0.0) { const double2 p = st[e]; er = p.x * alpha + pr * (1.0 - alpha); ei = p.y * alpha + pi * (1.0 - alpha); }
            sr += er; si += ei; st[e] = make_double2(er, ei);
          }
        double numerator;
        if (1 & type) { numerator = sr; if (numerator < 0.0) numerator = 0.0; } else numerator = hypot(sr, si);
        double denominator = 0.0;
        for (int i = 0; i < C; i++) {
          const double a2 = ta[i].x * ta[i].x + ta[i].y * ta[i].y;
          const size_t e = (size_t) (i * C + i) * F;
          double est = a2;
          if (alpha > 0.0) est = alpha * st[e].x + (1.0 - alpha) * a2;
          denominator += est; st[e] = make_double2(est, 0.0);
        }
        double W = (numerator / denominator) * (2.0 / ((double) C - 1.0));
        if (W >= 1.0) W = 1.0;
        if (W < 0.0001) W = 0.0001;
        return W;                                                  // (type 0 = NO_USE_POST_FILTER: the densities are updated, the caller does not filter)
      };
      dot(weff + ((size_t) target * F + f) * C, tr, ti);
      const double ur = tr, ui = ti;                               // before the post-filter: what the mirror bin of a full frame keeps
      const bool filt = zel && type != 0;
      if (apab) apabApply(target, weff, tr, ti);
      else { const double W = pfw(target); if (filt) { const double a = W * tr - 0.0 * ti, b = W * ti + 0.0 * tr; tr = a; ti = b; } }   // polar(W, 0) * y
      bool masked = false;
      const bool maskBin = useMask && (hbs || f >= 1);             // bin 0 is never masked (:2283), but calcInterferenceOutputs runs its post-filters too:
      if (useMask) {                                               // with mask type 1 that is the second update of the target's densities in this frame
        for (int s = 0; s < S; s++) {
          if (maskType == 0 && s == target) continue;
          double yr, yi;
          dot((maskType == 0 ? weff : wup) + ((size_t) s * F + f) * C, yr, yi);
          if (apab) apabApply(s, maskType == 0 ? weff : wup, yr, yi);
          else { const double W = pfw(s); if (filt) { const double a = W * yr - 0.0 * yi, b = W * yi + 0.0 * yr; yr = a; yi = b; } }
          const double p = yr * yr + yi * yi;
          if (s == target) tgtPow = p; else if (p > maxPow) maxPow = p;
        }
        if (maskType == 0) tgtPow = tr * tr + ti * ti;             // _interferenceOutputs[target] = the post-filtered output (:2064-2065)
        masked = maskBin && tgtPow < maxPow;
      }
      const bool mir = fullOut && f >= 1 && f < M2;                // bin M - f of a full frame: conj of the unfiltered value (:2026-2027) unless the mask strikes
      if (!maskBin) { Yu[(size_t) t * Fout + f] = make_float2((float) tr, (float) ti); if (mir) Yu[(size_t) t * Fout + (M - f)] = make_float2((float) ur, (float) -ui); }
      else if (!wide) {
        // binaryMasking (:2241-2319) for one bin: the averaged output is this bin's own recursion
        double nr = 0.0, ni = 0.0;
        if (avgFactor >= 0.0) { nr = avgOut[f].x * avgFactor; ni = avgOut[f].y * avgFactor; }
        if (masked) { tr = nr; ti = ni; if (avgFactor >= 0.0) avgOut[f] = make_double2(nr, ni); }
        else if (avgFactor >= 0.0) avgOut[f] = make_double2(avgOut[f].x * avgFactor + tr * (1.0 - avgFactor), avgOut[f].y * avgFactor + ti * (1.0 - avgFactor));
        Yu[(size_t) t * Fout + f] = make_float2((float) tr, (float) ti);
        if (mir) Yu[(size_t) t * Fout + (M - f)] = masked ? make_float2((float) tr, (float) ti) : make_float2((float) ur, (float) -ui);
      } else { outS[f] = make_double2(tr, ti); maskS[f] = masked ? 1 : 0; mirS[f] = make_double2(ur, -ui); }
    }
    if (wide) {
      // the mean over neighbouring bins reads values this frame's loop has already replaced below the bin and last frame's above it
      // (getMeanOfSubbandC on _avgOutput while it is being updated, :2212-2231): one thread, bins in order
      __syncthreads();
      if (tid == 0) {
        const int f0 = hbs ? 0 : 1, f1 = hbs ? M - 1 : M / 2, lim = hbs ? M : M / 2;
        for (int f = f0; f <= f1; f++) {
          int fs = f - fwidth / 2; if (fs < 1) fs = 1;
          int fe = f + fwidth / 2; if (fe >= lim) fe = lim - 1;
          double sr = 0.0, si = 0.0; unsigned cnt = 0;
          for (int i = fs; i <= fe; i++, cnt++) { sr += avgOut[i].x; si += avgOut[i].y; }
          const double nr = (sr / (double) cnt) * avgFactor, ni = (si / (double) cnt) * avgFactor;
          if (maskS[f]) { outS[f] = make_double2(nr, ni); avgOut[f] = make_double2(nr, ni); }
          else avgOut[f] = make_double2(avgOut[f].x * avgFactor + outS[f].x * (1.0 - avgFactor), avgOut[f].y * avgFactor + outS[f].y * (1.0 - avgFactor));
        }
      }
      __syncthreads();
      for (int f = tid; f < F; f += nthr) if (hbs || f >= 1) {
        Yu[(size_t) t * Fout + f] = make_float2((float) outS[f].x, (float) outS[f].y);
        if (fullOut && f < M2) { const double2 v = maskS[f] ? outS[f] : mirS[f]; Yu[(size_t) t * Fout + (M - f)] = make_float2((float) v.x, (float) v.y); }
      }
    }
  }
}

}  // namespace dsr

using namespace dsr;

struct dsr_mmi {
  int M = 0, C = 0, hbs = 0, target = 0, S = 2, pfType = 0, NC = 1; double alpha = 0.9;
  bool haveW = false, dirty = true;
  bool useMask = false; double avgFactor = -1.0; unsigned fwidth = 1, maskType = 0;
  std::vector<SrcW> src;
  DevBuf<double2> d_weff, d_wup, d_mani, d_csd, d_ta;
};

static void mmi_alloc(dsr_mmi& m, int NC)                // _allocBFWeight (:1124-1134): fresh, zeroed weight objects
{
  const size_t M = m.M, C = m.C, bs = C - NC;
  m.src.assign(m.S, SrcW()); m.NC = NC;
  for (auto& w : m.src) { w.wq.assign(M * C, zc(0, 0)); w.B.assign(M * C * bs, zc(0, 0)); w.wa.assign(M * bs, zc(0, 0)); w.wl.assign(M * C, zc(0, 0)); w.ta.assign(M * C, zc(0, 0)); }
  m.haveW = true; m.dirty = true;
}

static void mmi_mainlobe(dsr_mmi& m, SrcW& w, double fs, const double* delays, bool isGSC)     // calcMainlobe (:531-594)
{
  const int M = m.M, C = m.C, M2 = M / 2;
  if (m.hbs) {
    const float fshift = 0.5f;
    for (int f = 0; f < M2; f++)
      for (int c = 0; c < C; c++) {
        const double val = -2.0 * M_PI * (fshift + f) * fs * delays[c] / M;
        const zc a = gpolar(1.0, val), b = gpolar(1.0, -val);
        w.wq[(size_t) f * C + c] = zc(a.real() / C, a.imag() / C); w.wq[(size_t) (M - 1 - f) * C + c] = zc(b.real() / C, b.imag() / C);
      }
  } else {
    for (int c = 0; c < C; c++) { const zc a = gpolar(1.0, 0.0); w.wq[c] = zc(a.real() / C, a.imag() / C); }
    for (int f = 1; f < M2; f++)
      for (int c = 0; c < C; c++) {
        const double val = -2.0 * M_PI * f * delays[c] * fs / M;
        const zc a = gpolar(1.0, val), b = gpolar(1.0, -val);
        w.wq[(size_t) f * C + c] = zc(a.real() / C, a.imag() / C); w.wq[(size_t) (M - f) * C + c] = zc(b.real() / C, b.imag() / C);
      }
    for (int c = 0; c < C; c++) { const zc a = gpolar(1.0, -M_PI * fs * delays[c]); w.wq[(size_t) M2 * C + c] = zc(a.real() / C, a.imag() / C); }
  }
  w.ta = w.wq;                                                               // setTimeAlignment (:992-997)
  if (isGSC) for (int f = 0; f < M; f++) blocking_matrix_nc(&w.wq[(size_t) f * C], C, 1, &w.B[(size_t) f * C * (C - m.NC)]);
}

static void mmi_mainlobe_n(dsr_mmi& m, SrcW& w, double fs, const double* delaysT, const double* delaysIs, int NC)     // calcMainlobeN (:631-753), isGSC
{
  const int M = m.M, C = m.C, M2 = M / 2;
  std::vector<std::vector<zc>> pWj(NC - 1, std::vector<zc>(C)), pWjConj(NC - 1, std::vector<zc>(C));
  mmi_mainlobe(m, w, fs, delaysT, false);
  if (m.hbs) {
    const float fshift = 0.5f;
    for (int f = 0; f < M2; f++) {
      zc* vec = &w.wq[(size_t) f * C]; zc* vecConj = &w.wq[(size_t) (M - 1 - f) * C];
      for (int c = 0; c < C; c++) {
        vec[c] = zc(vec[c].real() * C, vec[c].imag() * C); vecConj[c] = zc(vecConj[c].real() * C, vecConj[c].imag() * C);
        for (int n = 0; n < NC - 1; n++) {
          const double valJ = -2.0 * M_PI * (fshift + f) * fs * delaysIs[(size_t) n * C + c] / M;
          pWj[n][c] = gpolar(1.0, valJ); pWjConj[n][c] = gpolar(1.0, -valJ);
        }
      }
      calc_null_beamformer(vec, pWj, C, NC); calc_null_beamformer(vecConj, pWjConj, C, NC);
    }
  } else {
    for (int c = 0; c < C; c++) w.wq[c] = zc(1.0 / C, 0.0);
    for (int f = 1; f < M2; f++) {
      zc* vec = &w.wq[(size_t) f * C];
      for (int c = 0; c < C; c++) {
        vec[c] = zc(vec[c].real() * C, vec[c].imag() * C);
        for (int n = 0; n < NC - 1; n++) pWj[n][c] = gpolar(1.0, -2.0 * M_PI * f * fs * delaysIs[(size_t) n * C + c] / M);
      }
      calc_null_beamformer(vec, pWj, C, NC);                                 // (the mirror bins keep their delay-and-sum vectors; they are never applied)
    }
    // bin M/2 as shipped (:718-731): each channel's entry is replaced by an interference phase term and the null beamformer is solved inside
    // the channel loop with the manifolds left from bin M/2 - 1
    zc* vec = &w.wq[(size_t) M2 * C];
    for (int c = 0; c < C; c++) {
      vec[c] = zc(vec[c].real() * C, vec[c].imag() * C);
      for (int n = 0; n < NC - 1; n++) { const zc a = gpolar(1.0, -M_PI * fs * delaysIs[(size_t) n * C + c]); vec[c] = zc(a.real() / C, a.imag() / C); }
      calc_null_beamformer(vec, pWj, C, NC);
    }
  }
  for (int f = 0; f < M; f++) blocking_matrix_nc(&w.wq[(size_t) f * C], C, NC, &w.B[(size_t) f * C * (C - NC)]);
}

static void mmi_update_wl(dsr_mmi& m, SrcW& w, unsigned f)      // wl = B wa (:761-799)
{
  const int C = m.C, bs = C - m.NC;
  for (int i = 0; i < C; i++) { zc acc(0, 0); for (int j = 0; j < bs; j++) acc += gmul(w.B[((size_t) f * C + i) * bs + j], w.wa[(size_t) f * bs + j]); w.wl[(size_t) f * C + i] = acc; }
}

dsr_status dsr_mmi_create(int fftLen, int chanN, int halfBandShift, int targetSourceX, int nSource, int pfType, double alpha, dsr_mmi** out)
{
  return guard([&] {
    if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (fftLen < 4 || (fftLen & 1) || chanN < 2 || nSource < 1 || targetSourceX < 0 || targetSourceX >= nSource) throw Error(DSR_E_PARAMETER, "SubbandMMI: fftLen %d, %d channels, source %d of %d", fftLen, chanN, targetSourceX, nSource);
    auto* m = new dsr_mmi(); m->M = fftLen; m->C = chanN; m->hbs = halfBandShift ? 1 : 0; m->target = targetSourceX; m->S = nSource; m->pfType = pfType; m->alpha = alpha;
    *out = m;
  });
}
void dsr_mmi_destroy(dsr_mmi* m) { delete m; }
int dsr_mmi_chan_n(const dsr_mmi* m) { return m ? m->C : 0; }
int dsr_mmi_fft_len(const dsr_mmi* m) { return m ? m->M : 0; }
int dsr_mmi_bins(const dsr_mmi* m) { return m ? (m->hbs ? m->M : m->M / 2 + 1) : 0; }
int dsr_mmi_out_bins(const dsr_mmi* m) { return m ? ((m->hbs || (m->pfType & 0x04)) ? m->M : m->M / 2 + 1) : 0; }

dsr_status dsr_mmi_use_binary_mask(dsr_mmi* m, double avgFactor, unsigned fwidth, unsigned type)
{
  return guard([&] {
    if (!m) throw Error(DSR_E_PARAMETER, "null argument");
    if (type > 1) throw Error(DSR_E_PARAMETER, "binary mask type %u", type);
    m->useMask = true; m->avgFactor = avgFactor; m->fwidth = fwidth; m->maskType = type;
  });
}

dsr_status dsr_mmi_calc_weights(dsr_mmi* m, double sampleRate, const double* delays)
{
  return guard([&] {
    if (!m || !delays) throw Error(DSR_E_PARAMETER, "null argument");
    mmi_alloc(*m, 1);
    for (int s = 0; s < m->S; s++) mmi_mainlobe(*m, m->src[s], sampleRate, delays + (size_t) s * m->C, true);
  });
}

dsr_status dsr_mmi_calc_weights_n(dsr_mmi* m, double sampleRate, const double* delays, unsigned NC)
{
  return guard([&] {
    if (!m || !delays) throw Error(DSR_E_PARAMETER, "null argument");
    if (NC < 2 || NC > (unsigned) m->C) throw Error(DSR_E_DIMENSION, "1 < the number of constraints %u <= the number of sensors %d.", NC, m->C);   // :633-635
    if (NC > (unsigned) m->S) throw Error(DSR_E_DIMENSION, "%u constraints need %u sources, there are %d", NC, NC, m->S);                  // rows of the delay matrix (:1800-1806)
    if (!m->haveW) mmi_alloc(*m, (int) NC);                                   // only when there are no weight objects yet (:1790-1791)
    if ((unsigned) m->NC != NC) throw Error(DSR_E_DIMENSION, "the weights were allocated for %d constraints, not %u", m->NC, NC);
    std::vector<double> delaysIs((size_t) (NC - 1) * m->C);
    for (int s = 0; s < m->S; s++) {
      for (unsigned srcY = 0, i = 0; i < NC - 1; srcY++) {
        if ((int) srcY == s) continue;
        for (int c = 0; c < m->C; c++) delaysIs[(size_t) i * m->C + c] = delays[(size_t) srcY * m->C + c];
        i++;
      }
      mmi_mainlobe_n(*m, m->src[s], sampleRate, delays + (size_t) s * m->C, delaysIs.data(), (int) NC);
    }
    m->dirty = true;
  });
}

dsr_status dsr_mmi_set_active_weights_f(dsr_mmi* m, unsigned fbinX, const double* packed, size_t rows, size_t cols, int option)
{
  return guard([&] {
    if (!m || !packed) throw Error(DSR_E_PARAMETER, "null argument");
    if (!m->haveW) throw Error(DSR_E_ERROR, "call calcWeightsX() once");                                                       // :1823-1826
    if (rows != (size_t) m->S) throw Error(DSR_E_ERROR, "The number of columns must be the number of sources %d", m->S);      // :1827-1830
    const int C = m->C, bs = C - m->NC, S = m->S;
    if (cols != (size_t) 2 * bs) throw Error(DSR_E_DIMENSION, "the size of an active weight vector must be %d but it is %zu", 2 * bs, cols);   // :764-766
    if (fbinX >= (unsigned) m->M) throw Error(DSR_E_DIMENSION, "Must be a frequency bin %u < the length of FFT %d", fbinX, m->M);
    for (int s = 0; s < S; s++) {
      for (int c = 0; c < bs; c++) m->src[s].wa[(size_t) fbinX * bs + c] = zc(packed[(size_t) s * 2 * bs + 2 * c], packed[(size_t) s * 2 * bs + 2 * c + 1]);
      mmi_update_wl(*m, m->src[s], fbinX);
    }
    std::vector<zc> Wl((size_t) S * C);
    for (int s = 0; s < S; s++) for (int c = 0; c < C; c++) Wl[(size_t) s * C + c] = std::conj(m->src[s].wl[(size_t) fbinX * C + c]);
    if (option == 1) scaling(Wl.data(), S, C, 1.0E-7f);
    for (int s = 0; s < S; s++) for (int c = 0; c < C; c++) m->src[s].wl[(size_t) fbinX * C + c] = std::conj(Wl[(size_t) s * C + c]);
    m->dirty = true;
  });
}

dsr_status dsr_mmi_set_hi_active_weights_f(dsr_mmi* m, unsigned fbinX, const double* pkdWa, size_t nWa, const double* pkdwb, size_t nWb, int option)
{
  return guard([&] {
    if (!m || !pkdWa || !pkdwb) throw Error(DSR_E_PARAMETER, "null argument");
    if (!m->haveW) throw Error(DSR_E_ERROR, "call calcWeightsX() once");
    const int C = m->C, bs = C - m->NC, S = m->S;
    if (nWa != (size_t) 2 * S * bs * S) throw Error(DSR_E_ERROR, "The size of the 2nd arg must be 2 * %d * %d * %d", S, bs, S);   // :1899-1902
    if (nWb != (size_t) 2 * S * S) throw Error(DSR_E_ERROR, "The size of the 3rd arg must be 2 * %d * %d", S, S);
    if (fbinX >= (unsigned) m->M) throw Error(DSR_E_DIMENSION, "Must be a frequency bin %u < the length of FFT %d", fbinX, m->M);
    std::vector<zc> Wc((size_t) S * S);
    for (int k = 0; k < S * S; k++) Wc[k] = std::conj(zc(pkdwb[2 * k], pkdwb[2 * k + 1]));
    if (option == 1) scaling(Wc.data(), S, S, 1.0E-7f);
    for (int s = 0; s < S; s++) {
      SrcW& w = m->src[s];
      for (int c = 0; c < bs; c++) {
        zc acc(0, 0);
        for (int y = 0; y < S; y++) { const size_t i = ((size_t) s * bs + c) * S + y; acc += gmul(zc(pkdWa[2 * i], pkdWa[2 * i + 1]), std::conj(Wc[(size_t) s * S + y])); }
        w.wa[(size_t) fbinX * bs + c] = acc;
      }
      mmi_update_wl(*m, w, fbinX);
    }
    m->dirty = true;
  });
}

dsr_status dsr_mmi_get(const dsr_mmi* m, int srcX, int kind, double* out, size_t outDoubles)
{
  return guard([&] {
    if (!m || !out) throw Error(DSR_E_PARAMETER, "null argument");
    if (!m->haveW) throw Error(DSR_E_ERROR, "call calcWeightsX() once");
    if (srcX < 0 || srcX >= m->S) throw Error(DSR_E_INDEX, "source %d of %d", srcX, m->S);
    const SrcW& w = m->src[srcX];
    const std::vector<zc>* p = kind == 0 ? &w.wq : kind == 1 ? &w.wl : kind == 2 ? &w.B : kind == 3 ? &w.ta : kind == 4 ? &w.wa : nullptr;
    if (!p) throw Error(DSR_E_PARAMETER, "kind %d", kind);
    if (outDoubles < 2 * p->size()) throw Error(DSR_E_DIMENSION, "buffer of %zu doubles for %zu", outDoubles, 2 * p->size());
    for (size_t k = 0; k < p->size(); k++) { out[2 * k] = (*p)[k].real(); out[2 * k + 1] = (*p)[k].imag(); }
  });
}

dsr_status dsr_mmi_apply(dsr_mmi* m, const float* X, const int32_t* nframes_dev, int U, int Tmax, float* Y, void* stream)
{
  return guard([&] {
    if (!m || !X || !Y || !nframes_dev) throw Error(DSR_E_PARAMETER, "null argument");
    if (!m->haveW) throw Error(DSR_E_ERROR, "call calcWeightsX() once");                 // :1984-1987
    if (U <= 0 || Tmax <= 0) return;
    hipStream_t st = (hipStream_t) stream;
    const int M = m->M, C = m->C, S = m->S, F = m->hbs ? M : M / 2 + 1;
    if (m->dirty) {
      std::vector<double2> weff((size_t) S * F * C), wup((size_t) S * F * C), mani((size_t) S * F * C);
      for (int s = 0; s < S; s++)
        for (int f = 0; f < F; f++)
          for (int c = 0; c < C; c++) {
            const size_t k = ((size_t) s * F + f) * C + c; const zc wq = m->src[s].wq[(size_t) f * C + c], wl = m->src[s].wl[(size_t) f * C + c];
            const zc e = (!m->hbs && f == 0) ? wq : wq - wl;                            // bin 0: the quiescent vector alone (:2012-2016)
            weff[k] = make_double2(e.real(), e.imag()); wup[k] = make_double2(wq.real(), wq.imag());
            const zc d = ((m->pfType & 0x08) && !(m->pfType & 0x04)) ? wq : m->src[s].ta[(size_t) f * C + c];     // TYPE_ZELINSKI2: the beamformer's own vector (:2052-2053); APAB: always the manifold
            mani[k] = make_double2(d.real(), d.imag());
          }
      m->d_weff.upload(weff); m->d_wup.upload(wup); m->d_mani.upload(mani); m->dirty = false;
    }
    const bool zel = !(m->pfType & 0x04) && ((m->pfType & 0x01) || (m->pfType & 0x02));
    m->d_csd.reserve(zel ? (size_t) S * U * C * C * F : 1);
    if (C > 16) m->d_ta.reserve((size_t) U * F * C);
    const size_t lds = (size_t) F * (3 * sizeof(double2) + sizeof(int));
    if (lds > 150 * 1024) throw Error(DSR_E_DIMENSION, "SubbandMMI: %d bins need %zu bytes of LDS", F, lds);
#define MMI_ARGS (const float2*) X, nframes_dev, m->d_weff.p, m->d_wup.p, m->d_mani.p, m->d_csd.p, m->d_ta.p, (float2*) Y, U, C, Tmax, F, S, m->target, m->hbs, \
                 m->pfType, m->alpha, m->useMask ? 1 : 0, (int) m->maskType, m->avgFactor, (int) m->fwidth, M, dsr_mmi_out_bins(m)
    if (C <= 16) {
      DSR_HIP(hipFuncSetAttribute((const void*) k_mmi<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      hipLaunchKernelGGL(k_mmi<16>, dim3(U), dim3(256), lds, st, MMI_ARGS);
    } else {
      DSR_HIP(hipFuncSetAttribute((const void*) k_mmi<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      hipLaunchKernelGGL(k_mmi<0>, dim3(U), dim3(256), lds, st, MMI_ARGS);
    }
#undef MMI_ARGS
    DSR_HIP(hipGetLastError());
  });
}
