// csrc/k_gmm.hip -- diagonal-covariance GMM scoring (costs = negative log-likelihoods).
//
// Replaces CodebookBasic::_scoreOpt (asr/gaussian/codebookBasic.cc:431-554), ::_scoreAll (:645-766)
// and DistribBasic::_score (asr/gaussian/distribBasic.h:110-114) for whole frame batches: the
// reference scores one (codebook, frame) at a time behind a per-frame cache; here every
// (frame, codebook) of a batch is scored in one launch.
//
//   mode 0  k_gmm_exact : thread per frame, Gaussians broadcast from LDS; the Mahalanobis sum is
//           accumulated in fp32 in the reference's order (d ascending, (mu-x)^2*iv, no FMA) so the
//           score and argmin are bit-identical.  The reference's early exit is omitted: partial sums
//           of non-negative terms are monotone, so it cannot change the strict-'<' argmin.
//   mode 1  k_gmm_all   : fp64 per-Gaussian distances + log-sum with the -100 exponent floor.
//   mode 2  k_gmm_mfma  : frame x Gaussian contraction on v_mfma_f32_32x32x2_f32 in the expanded
//           form  sum_d iv x^2 - 2 mu iv x + (mu^2 iv) ; the per-codebook candidates that could still be
//           the fp32-ordered minimum (rigorous rounding bound) are re-scored in reference order, so
//           the output bits equal mode 0.
// This translation unit is compiled with -ffp-contract=off.
#include "common.h"
#include "gmm_model.h"
#include <cmath>
#include <string>

namespace dsr {


// ------------------------------------------------------------------------------------------------
// mode 0: exact nearest-Gaussian.  Block = 256 threads = 256 frames; loop over codebooks, Gaussian
// parameters of a codebook chunk staged in LDS and read as wave-wide broadcasts.
static constexpr int kTW = 16;                            // codebooks per score tile of k_gmm_exact
template <int DP>
__global__ __launch_bounds__(256) void k_gmm_exact(const float* __restrict__ x, long N, int D, int K, const int* __restrict__ off,
                                                   const float* __restrict__ mean, const float* __restrict__ ivar,
                                                   const float* __restrict__ cst, const float* __restrict__ val,
                                                   const float* __restrict__ scale, float* __restrict__ score,
                                                   unsigned char* __restrict__ argmin, int chunkG, int finish)
{
  // finish 0: _scoreOpt's 0.5 (min + 2 val) x codebook scale (codebookBasic.cc:533-549); 1 / 2: CodebookBasic::logLhood's own rounding,
  // 0.5 min + val[argmin] / 0.5 min when val is NULL, no scale (:600-606)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sm = reinterpret_cast<float*>(smem);            // [chunkG][DP] means
  float* sv = sm + (size_t) chunkG * DP;                  // [chunkG][DP] inverse variances
  // scores of a chunk's codebooks (at most kTW) wait in an LDS tile [256 frames][kTW + 1] and leave as 64-byte row segments: a thread
  // storing its own frame's scores one by one would write 4 bytes per lane at a stride of K floats, every store a partial line
  float* tile = sv + (size_t) chunkG * DP;
  unsigned char* tileI = reinterpret_cast<unsigned char*>(tile + 256 * (kTW + 1));     // [256][kTW + 4] nearest-Gaussian indices, same way
  const long n = (long) blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = n < N;
  float xr[DP];
#pragma unroll
  for (int d = 0; d < DP; d++) xr[d] = (live && d < D) ? x[n * D + d] : 0.0f;

  int k = 0;
  while (k < K) {
    // take as many whole codebooks as fit in the chunk
    const int g0 = off[k]; int k1 = k;
    while (k1 < K && k1 - k < kTW && off[k1 + 1] - g0 <= chunkG) k1++;
    const int ng = off[k1] - g0;
    __syncthreads();
    for (int i = threadIdx.x; i < ng * DP; i += blockDim.x) { sm[i] = mean[(long) g0 * DP + i]; sv[i] = ivar[(long) g0 * DP + i]; }
    __syncthreads();
    for (int kk = k; kk < k1; kk++) {
      const int a = off[kk] - g0, b = off[kk + 1] - g0;
      float minDist = 1E20f; int minIdx = 0;
      for (int g = a; g < b; g++) {
        float dist = cst[g0 + g];
        const float* rv = sm + (size_t) g * DP; const float* cv = sv + (size_t) g * DP;
#pragma unroll
        for (int d = 0; d < DP; d++) {
          // padded dimensions hold mu = x = iv = 0 and add exactly +0
          { const float diff = __fsub_rn(rv[d], xr[d]); dist = __fadd_rn(dist, __fmul_rn(__fmul_rn(diff, diff), cv[d])); }
        }
        if (dist < minDist) { minDist = dist; minIdx = g - a; }
      }
      if (live) {
        float sc;
        if (finish == 0) {
          sc = (float) (0.5 * (double) __fadd_rn(minDist, __fmul_rn(2.0f, val[off[kk] + minIdx])));
          const float s = scale[kk]; if (s != 1.0f) sc = __fmul_rn(sc, s);
        } else {
          sc = __fmul_rn(minDist, 0.5f);                                   // "minDistSum *= 0.5" (exact), then "+= val[minDistIdx]" in float
          if (finish == 1) sc = __fadd_rn(sc, val[off[kk] + minIdx]);
        }
        tile[threadIdx.x * (kTW + 1) + (kk - k)] = sc;
        if (argmin) tileI[threadIdx.x * (kTW + 4) + (kk - k)] = (unsigned char) minIdx;
      }
    }
    __syncthreads();
    {
      const int nk = k1 - k, col = threadIdx.x & (kTW - 1), r0 = threadIdx.x / kTW;
      const long nb = (long) blockIdx.x * blockDim.x;
      if (col < nk)
        for (int r = r0; r < 256; r += 256 / kTW) if (nb + r < N) {
          score[(nb + r) * K + k + col] = tile[r * (kTW + 1) + col];
          if (argmin) argmin[(nb + r) * K + k + col] = tileI[r * (kTW + 4) + col];
        }
    }
    k = k1;
  }
}

// mode 1: full mixture (codebookBasic.cc:710-765)
__global__ __launch_bounds__(256) void k_gmm_all(const float* __restrict__ x, long N, int D, int DP, int K, const int* __restrict__ off,
                                                 const float* __restrict__ mean, const float* __restrict__ ivar,
                                                 const float* __restrict__ cst, const float* __restrict__ val,
                                                 const float* __restrict__ scale, float* __restrict__ score)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * K) return;
  const long n = idx / K; const int k = (int) (idx - n * K);
  const float* xp = x + n * D;
  const int a = off[k], b = off[k + 1], R = b - a; const float sc = scale[k];
  double minlog = 1E20;
  for (int g = a; g < b; g++) {
    double dist = (double) cst[g];
    for (int d = 0; d < D; d++) { const double diff = (double) __fsub_rn(mean[(long) g * DP + d], xp[d]); dist = __dadd_rn(dist, __dmul_rn(__dmul_rn(diff, diff), (double) ivar[(long) g * DP + d])); }
    const float ld = (float) (0.5 * dist);
    if ((double) ld < minlog) minlog = (double) ld;
  }
  float res;
  if (R == 1) {
    double dist = (double) cst[a];
    for (int d = 0; d < D; d++) { const double diff = (double) __fsub_rn(mean[(long) a * DP + d], xp[d]); dist = __dadd_rn(dist, __dmul_rn(__dmul_rn(diff, diff), (double) ivar[(long) a * DP + d])); }
    res = __fmul_rn(sc, __fadd_rn((float) (0.5 * dist), val[a]));
  } else {
    double s = 0.0;
    for (int g = a; g < b; g++) {        // second pass recomputes the cached distances
      double dist = (double) cst[g];
      for (int d = 0; d < D; d++) { const double diff = (double) __fsub_rn(mean[(long) g * DP + d], xp[d]); dist = __dadd_rn(dist, __dmul_rn(__dmul_rn(diff, diff), (double) ivar[(long) g * DP + d])); }
      const float ld = (float) (0.5 * dist);
      const double e = __dmul_rn((double) sc, __dsub_rn(minlog, (double) ld));
      if (e > -100.0) s = __dadd_rn(s, exp(__dsub_rn(e, (double) __fmul_rn(sc, val[g]))));
    }
    res = (float) __dsub_rn(__dmul_rn((double) sc, minlog), log(s));
  }
  score[idx] = res;
}

}  // namespace dsr

using namespace dsr;
struct dsr_gmm : GmmModel {};

namespace dsr {
void gmm_finish(GmmModel& m)
{
  m.off.assign(m.K + 1, 0); m.maxRef = 0;
  for (int k = 0; k < m.K; k++) {
    if (m.refN[k] < 1 || m.refN[k] > 256) throw Error(DSR_E_DIMENSION, "codebook %d has %d Gaussians (1..256, CBX is unsigned char)", k, m.refN[k]);
    m.off[k + 1] = m.off[k] + m.refN[k]; if (m.refN[k] > m.maxRef) m.maxRef = m.refN[k];
  }
  m.G = m.off[m.K];
  m.Dp = m.D <= 16 ? 16 : m.D <= 40 ? 40 : m.D <= 64 ? 64 : 128;   // padded row length (exact kernel templates)
  m.pi.assign(m.K, (float) (log(2.0 * M_PI) * m.D));           // float _pi, codebookBasic.cc:170
  if (m.scale.empty()) m.scale.assign(m.K, 1.0f);
  if (m.count.empty()) m.count.assign(m.G, 1.0f);
  std::vector<float> pm((size_t) m.G * m.Dp, 0.f), pv((size_t) m.G * m.Dp, 0.f), cst(m.G);
  for (int k = 0; k < m.K; k++) for (int g = m.off[k]; g < m.off[k + 1]; g++) cst[g] = m.pi[k] + m.det[g];   // float + float (:481)
  for (int g = 0; g < m.G; g++) for (int d = 0; d < m.D; d++) { pm[(size_t) g * m.Dp + d] = m.mean[(size_t) g * m.D + d]; pv[(size_t) g * m.Dp + d] = m.ivar[(size_t) g * m.D + d]; }
  require_device();
  m.d_off.upload(m.off); m.d_mean.upload(pm); m.d_ivar.upload(pv); m.d_cst.upload(cst); m.d_val.upload(m.val); m.d_scale.upload(m.scale);
  m.mfmaReady = false;
}
void gmm_score_mfma(GmmModel& m, const float* x, long N, float* score, unsigned char* argmin, hipStream_t st);   // k_gmm_mfma.hip
}

// ---- big-endian model files (btk/common/mach_ind_io.cc:176-510) ----
namespace {
struct BE {
  FILE* fp;
  int i32() { unsigned char b[4] = {0,0,0,0}; if (fread(b, 1, 4, fp) != 4) throw Error(DSR_E_IO, "premature end of file"); return (int) ((unsigned) b[0] << 24 | (unsigned) b[1] << 16 | (unsigned) b[2] << 8 | (unsigned) b[3]); }
  float f32() { int i = i32(); float f; memcpy(&f, &i, 4); return f; }
  short i16() { unsigned char b[2] = {0,0}; if (fread(b, 1, 2, fp) != 2) throw Error(DSR_E_IO, "premature end of file"); return (short) ((unsigned) b[0] << 8 | (unsigned) b[1]); }
  std::string str() { short len = i16(); std::string s((size_t) len + 1, '\0'); if (fread(&s[0], (size_t) len + 1, 1, fp) != 1) throw Error(DSR_E_IO, "premature end of file"); s.resize(len); return s; }
  void w32(int v) { unsigned u = (unsigned) v; unsigned char b[4] = { (unsigned char)(u >> 24), (unsigned char)(u >> 16), (unsigned char)(u >> 8), (unsigned char) u }; fwrite(b, 1, 4, fp); }
  void wf(float f) { int i; memcpy(&i, &f, 4); w32(i); }
  void w16(short v) { unsigned short u = (unsigned short) v; unsigned char b[2] = { (unsigned char)(u >> 8), (unsigned char) u }; fwrite(b, 1, 2, fp); }
  void wstr(const std::string& s) { w16((short) s.size()); fwrite(s.c_str(), s.size() + 1, 1, fp); }
};
}

extern "C" {

dsr_status dsr_gmm_create(int K, int D, const int32_t* refN, const float* mean, const float* ivar, const float* det,
                          const float* val, const float* scale, dsr_gmm** out)
{
  return guard([&] {
    if (!out || !refN || !mean || !ivar || !det || !val) throw Error(DSR_E_PARAMETER, "null argument");
    if (K < 1 || D < 1 || D > 128) throw Error(DSR_E_DIMENSION, "bad K=%d dimN=%d (dimN <= 128)", K, D);
    dsr_gmm* m = new dsr_gmm(); m->K = K; m->D = D; m->refN.assign(refN, refN + K);
    size_t G = 0; for (int k = 0; k < K; k++) G += (size_t) refN[k];
    m->mean.assign(mean, mean + G * D); m->ivar.assign(ivar, ivar + G * D); m->det.assign(det, det + G); m->val.assign(val, val + G);
    if (scale) m->scale.assign(scale, scale + K);
    for (int k = 0; k < K; k++) { m->cbNames.push_back("cb" + std::to_string(k)); m->dsNames.push_back("ds" + std::to_string(k)); }
    try { gmm_finish(*m); } catch (...) { delete m; throw; }
    *out = m;
  });
}

dsr_status dsr_gmm_load(const char* cbFile, const char* dsFile, dsr_gmm** out)
{
  return guard([&] {
    if (!cbFile || !dsFile || !out) throw Error(DSR_E_PARAMETER, "null argument");
    dsr_gmm* m = new dsr_gmm();
    FILE* fp = fopen(cbFile, "rb"); if (!fp) { delete m; throw Error(DSR_E_IO, "Could not open codebook file %s.", cbFile); }
    try {
      BE r{fp};
      const int first = r.i32();
      if (first != 64207531) {
        // the older (Janus) set format (CodebookSetBasic::load :934-957, CodebookBasic::loadOld :311-350; written by save(fp, janusFormat = true)
        // :352-383,962-983): codebook count, then per codebook name, refN, dimN, covariance type -- -1 = a count and a type per Gaussian -- and per
        // Gaussian [count] mean [type] inverse variances + determinant; no marker.  A negative count is the compressed mode the reference refuses.
        int cbN = first;
        if (cbN < 0) throw Error(DSR_E_IO, "Mode not supported.");
        m->K = cbN;
        for (int k = 0; k < m->K; k++) {
          m->cbNames.push_back(r.str());
          const int refN = r.i32(), dimN = r.i32(), lcov = r.i32();
          if (k == 0) m->D = dimN; else if (dimN != m->D) throw Error(DSR_E_DIMENSION, "codebooks of different dimension (%d vs %d)", dimN, m->D);
          m->refN.push_back(refN);
          for (int i = 0; i < refN; i++) {
            m->count.push_back(lcov == -1 ? r.f32() : 0.0f);
            for (int d = 0; d < dimN; d++) m->mean.push_back(r.f32());
            const int thisCov = lcov == -1 ? r.i32() : lcov;
            if (thisCov != 2 /*COV_DIAGONAL*/) throw Error(DSR_E_IO, "Wrong covariance type.");
            for (int d = 0; d < dimN; d++) m->ivar.push_back(r.f32());
            m->det.push_back(r.f32());
          }
        }
      }
      const int cb0 = first == 64207531 ? r.i32() : 0, cbN = first == 64207531 ? r.i32() : 0; if (first == 64207531) m->K = cbN - cb0;
      for (int k = 0; first == 64207531 && k < m->K; k++) {
        m->cbNames.push_back(r.str());                               // CodebookBasic::load :258-309
        const int refN = r.i32(), dimN = r.i32(), orgDimN = r.i32(), nSub = r.i32(); (void) r.i32();
        const int regP = r.i32(), descP = r.i32();
        if (k == 0) m->D = dimN; else if (dimN != m->D) throw Error(DSR_E_DIMENSION, "codebooks of different dimension (%d vs %d)", dimN, m->D);
        m->refN.push_back(refN);
        for (int i = 0; i < refN; i++) {
          m->count.push_back(r.f32());
          for (int d = 0; d < orgDimN; d++) { const float v = r.f32(); if (d < dimN) m->mean.push_back(v); }
          for (int d = 0; d < dimN; d++) m->ivar.push_back(r.f32());
          m->det.push_back(r.f32());
        }
        if (regP) for (int i = 0; i < refN; i++) { const int n = (unsigned short) r.i16(); for (int c = 0; c < n; c++) r.i16(); }
        if (descP) for (int i = 0; i < refN; i++) for (int b = 0; b < nSub; b++) r.i16();
        if (r.i32() != 123456789) throw Error(DSR_E_ERROR, "CheckMarker: Marker expected in codebook file.");
      }
      fclose(fp); fp = nullptr;
      fp = fopen(dsFile, "rb"); if (!fp) throw Error(DSR_E_IO, "Could not open distribution set file %s.", dsFile);
      BE q{fp};
      const int n = q.i32();                                          // DistribSetBasic::load, distribBasic.cc:254-285
      if (n != m->K) throw Error(DSR_E_CONSISTENCY, "%d distributions for %d codebooks: only 1:1 models are supported", n, m->K);
      m->val.clear();
      for (int i = 0; i < n; i++) {
        m->dsNames.push_back(q.str()); const std::string cbn = q.str();
        if (cbn != m->cbNames[i]) throw Error(DSR_E_ERROR, "Codebook names (%s vs. %s) do not match.", m->cbNames[i].c_str(), cbn.c_str());
        int rn = q.i32(); if (rn < 0) { rn = -rn; (void) q.f32(); }
        if (rn != m->refN[i]) throw Error(DSR_E_ERROR, "Distribution/codebook size mismatch: %s/%s %d/%d (not loaded!)", m->dsNames[i].c_str(), cbn.c_str(), rn, m->refN[i]);
        for (int j = 0; j < rn; j++) m->val.push_back(q.f32());
      }
      fclose(fp); fp = nullptr;
      gmm_finish(*m);
    } catch (...) { if (fp) fclose(fp); delete m; throw; }
    *out = m;
  });
}

// CodebookSetBasic::save(filename, janusFormat = true) (codebookBasic.cc:352-383,962-983): count, then name, refN, dimN, covariance type and per
// Gaussian mean, inverse variances, determinant -- no magic, no counts, no marker; the distribution file is the same in both formats
dsr_status dsr_gmm_save(const dsr_gmm* m, const char* cbFile, const char* dsFile);
dsr_status dsr_gmm_save_janus(const dsr_gmm* m, const char* cbFile, const char* dsFile)
{
  return guard([&] {
    if (!m || !cbFile) throw Error(DSR_E_PARAMETER, "null argument");
    FILE* fp = fopen(cbFile, "wb"); if (!fp) throw Error(DSR_E_IO, "Could not open codebook file %s.", cbFile);
    BE w{fp};
    w.w32(m->K);
    for (int k = 0; k < m->K; k++) {
      w.wstr(m->cbNames[k]); w.w32(m->refN[k]); w.w32(m->D); w.w32(2 /*COV_DIAGONAL*/);
      for (int g = m->off[k]; g < m->off[k + 1]; g++) {
        for (int d = 0; d < m->D; d++) w.wf(m->mean[(size_t) g * m->D + d]);
        for (int d = 0; d < m->D; d++) w.wf(m->ivar[(size_t) g * m->D + d]);
        w.wf(m->det[g]);
      }
    }
    fclose(fp);
    if (dsFile && *dsFile) { const dsr_status s = dsr_gmm_save(m, "/dev/null", dsFile); if (s) throw Error(s, "%s", dsr_last_error()); }
  });
}
dsr_status dsr_gmm_save(const dsr_gmm* m, const char* cbFile, const char* dsFile)
{
  return guard([&] {
    if (!m || !cbFile || !dsFile) throw Error(DSR_E_PARAMETER, "null argument");
    FILE* fp = fopen(cbFile, "wb"); if (!fp) throw Error(DSR_E_IO, "Could not open codebook file %s.", cbFile);
    BE w{fp};
    w.w32(64207531); w.w32(0); w.w32(m->K);                          // CodebookSetBasic::save :962-983
    for (int k = 0; k < m->K; k++) {
      w.wstr(m->cbNames[k]); w.w32(m->refN[k]); w.w32(m->D); w.w32(m->D); w.w32(1); w.w32(2 /*COV_DIAGONAL*/); w.w32(0); w.w32(0);
      for (int g = m->off[k]; g < m->off[k + 1]; g++) {
        w.wf(m->count[g]);
        for (int d = 0; d < m->D; d++) w.wf(m->mean[(size_t) g * m->D + d]);
        for (int d = 0; d < m->D; d++) w.wf(m->ivar[(size_t) g * m->D + d]);
        w.wf(m->det[g]);
      }
      w.w32(123456789);
    }
    fclose(fp);
    fp = fopen(dsFile, "wb"); if (!fp) throw Error(DSR_E_IO, "Could not open distribution set file %s.", dsFile);
    BE q{fp};
    q.w32(m->K);                                                      // DistribSetBasic::save / DistribBasic::save
    for (int k = 0; k < m->K; k++) {
      q.wstr(m->dsNames[k]); q.wstr(m->cbNames[k]); q.w32(-m->refN[k]); q.wf(0.0f);
      for (int g = m->off[k]; g < m->off[k + 1]; g++) q.wf(m->val[g]);
    }
    fclose(fp);
  });
}

void dsr_gmm_destroy(dsr_gmm* m) { delete m; }
int dsr_gmm_num_dists(const dsr_gmm* m) { return m->K; }
int dsr_gmm_dim(const dsr_gmm* m) { return m->D; }
// DistribSet::find(name) / index(key) (asr/gaussian/distribBasic.h:183-190: List lookup, jkey_error when absent) and the names of the set
const char* dsr_gmm_dist_name(const dsr_gmm* g, int distX) { return (g && distX >= 0 && distX < g->K) ? g->dsNames[distX].c_str() : ""; }
const char* dsr_gmm_codebook_name(const dsr_gmm* g, int cbX) { return (g && cbX >= 0 && cbX < g->K) ? g->cbNames[cbX].c_str() : ""; }
dsr_status dsr_gmm_find_dist(const dsr_gmm* g, const char* name, int* distX)
{
  return guard([&] {
    if (!g || !name || !distX) throw Error(DSR_E_PARAMETER, "null argument");
    for (int k = 0; k < g->K; k++) if (g->dsNames[k] == name) { *distX = k; return; }
    throw Error(DSR_E_KEY, "Could not find key %s in list Distribution Set", name);
  });
}

static dsr_status gmm_score_impl(dsr_gmm* m, const float* x, int64_t N, int mode, int finish, float* score, uint8_t* argmin, void* stream);
dsr_status dsr_gmm_score(dsr_gmm* m, const float* x, int64_t N, int mode, float* score, uint8_t* argmin, void* stream)
{ return gmm_score_impl(m, x, N, mode, 0, score, argmin, stream); }
// CodebookBasic::logLhood(frame, val) (codebookBasic.cc:557-609) for every (frame, codebook) of a batch: the nearest Gaussian as _scoreOpt finds it
// (same accumulation order, strict '<'; the early exit cannot change it) but finished as that method does: 0.5 * min, + val[argmin] when useVal
dsr_status dsr_gmm_log_lhood(dsr_gmm* m, const float* x, int64_t N, int useVal, float* score, uint8_t* argmin, void* stream)
{ return gmm_score_impl(m, x, N, 0, useVal ? 1 : 2, score, argmin, stream); }
static dsr_status gmm_score_impl(dsr_gmm* m, const float* x, int64_t N, int mode, int finish, float* score, uint8_t* argmin, void* stream)
{
  return guard([&] {
    if (!m || !x || !score) throw Error(DSR_E_PARAMETER, "null argument");
    if (N <= 0) return;
    hipStream_t st = (hipStream_t) stream;
    if (mode == 0) {
      // (20 KB of parameters = 16 codebooks of 4 Gaussians at 40 dims = one score tile; A/B 24 / 20 / 16 KB: 16.4 / 14.9 / 15.7 ms)
      int chunkG = (getenv("DSR_GMM_CHUNKKB") ? atoi(getenv("DSR_GMM_CHUNKKB")) : 20) * 1024 / (2 * m->Dp * 4); if (chunkG < m->maxRef) chunkG = m->maxRef;
      const size_t lds = (size_t) chunkG * m->Dp * 2 * sizeof(float) + (size_t) 256 * (kTW + 1) * sizeof(float) + (argmin ? (size_t) 256 * (kTW + 4) : 0);     // (the index tile only when indices are asked for: it costs a workgroup per CU)
      if (lds > 160 * 1024) throw Error(DSR_E_DIMENSION, "codebook too large for LDS staging");
      dim3 grid(cdiv(N, 256));
#define LAUNCH(DPV) { DSR_HIP(hipFuncSetAttribute((const void*) k_gmm_exact<DPV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
      hipLaunchKernelGGL(k_gmm_exact<DPV>, grid, dim3(256), lds, st, x, (long) N, m->D, m->K, m->d_off.p, m->d_mean.p, m->d_ivar.p, \
                         m->d_cst.p, m->d_val.p, m->d_scale.p, score, argmin, chunkG, finish); }
      if (m->Dp == 16) LAUNCH(16) else if (m->Dp == 40) LAUNCH(40) else if (m->Dp == 64) LAUNCH(64) else LAUNCH(128)
#undef LAUNCH
      DSR_HIP(hipGetLastError());
    } else if (mode == 1) {
      hipLaunchKernelGGL(k_gmm_all, dim3(cdiv(N * m->K, 256)), dim3(256), 0, st, x, (long) N, m->D, m->Dp, m->K, m->d_off.p, m->d_mean.p,
                         m->d_ivar.p, m->d_cst.p, m->d_val.p, m->d_scale.p, score);
      DSR_HIP(hipGetLastError());
    } else if (mode == 2) {
      gmm_score_mfma(*m, x, (long) N, score, argmin, st);
    } else throw Error(DSR_E_PARAMETER, "unknown scoring mode %d", mode);
  });
}

}  // extern "C"
