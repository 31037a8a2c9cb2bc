// csrc/k_lpc.hip -- LPC / MVDR spectral-envelope features, batched over frames.
//
// Replaces (btk/feature): BaseFeature::fftPower (lpc.cc:44-63), WarpFeature::autoCorrelation (lpc.cc:80-139),
// BurgFeature::autoCorrelation (lpc.cc:158-207), MVDRFeature<>::next (lpc.h:134-195) and LPCFeature<>::next
// (lpc.h:291-331), i.e. the operators WarpMVDRFeature, BurgMVDRFeature, WarpLPCFeature and BurgLPCFeature.
//
// The recursions of one frame are sequential and in fp32 (warped all-pass chain, Levinson-Durbin, Burg lattice); they
// are kept in the reference's order so that results agree to the last bit, and the parallelism comes from the frames:
// one thread per frame, every per-frame array stored [index][frame] so that a wave's accesses are contiguous.
// The spectrum of the (short) coefficient sequence is a direct fp64 DFT on the reference's 2^ceil(log2 dim)-point grid,
// of which -- as in the reference -- the first dim/2+1 bins are the output.  Compiled with -ffp-contract=off.
#include "common.h"
#include <cmath>

namespace dsr {

__global__ void k_lpc_transpose(const float* __restrict__ X, int Tc, int dim, float* __restrict__ XT)
{
  __shared__ float tile[32][33];
  const int j0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) { const int t = t0 + r, j = j0 + threadIdx.x; tile[r][threadIdx.x] = (t < Tc && j < dim) ? X[(size_t) t * dim + j] : 0.f; }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) { const int j = j0 + r, t = t0 + threadIdx.x; if (t < Tc && j < dim) XT[(size_t) j * Tc + t] = tile[threadIdx.x][r]; }
}

// WarpFeature::autoCorrelation (lpc.cc:80-139).  Scratch (all [index][Tc]): WX dim, R order+1, A0/A1 order+1.
__global__ void k_lpc_warp(const float* __restrict__ XT, int Tc, int dim, int order, float warp, float* __restrict__ WX,
                           float* __restrict__ R, float* __restrict__ A0, float* __restrict__ A1, float* __restrict__ LP, float* __restrict__ E0)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Tc) return;
  const size_t S = (size_t) Tc;
  float sum = 0.0f;
  for (int j = 0; j < dim; j++) { const float x = XT[j * S + t]; sum += x * x; WX[j * S + t] = x; }
  R[t] = sum;
  for (int i = 1; i <= order; i++) {
    // one all-pass stage in place: new[j] = warp (new[j-1] - old[j]) + old[j-1], new[0] = -warp old[0]
    float prevOld = 0.0f, prevNew = 0.0f; sum = 0.0f;
    for (int j = 0; j < dim; j++) {
      const float old = WX[j * S + t];
      const float nw = (j == 0) ? -warp * old : warp * (prevNew - old) + prevOld;
      WX[j * S + t] = nw; prevOld = old; prevNew = nw;
      sum += XT[j * S + t] * nw;
    }
    R[i * S + t] = sum;
  }
  float E = R[t];
  E0[t] = E;
  float* prev = A0; float* cur = A1;
  for (int i = 1; i <= order; i++) {
    float k = R[i * S + t];
    for (int j = 1; j < i; j++) k -= prev[j * S + t] * R[(i - j) * S + t];
    if (E != 0) k /= E; else k = 1000000000;
    cur[i * S + t] = k;
    for (int j = 1; j <= i - 1; j++) cur[j * S + t] = prev[j * S + t] - k * prev[(i - j) * S + t];
    E = (1 - k * k) * E;
    float* tmp = prev; prev = cur; cur = tmp;
  }
  LP[t] = 1.0f;
  for (int i = 1; i <= order; i++) LP[i * S + t] = -prev[i * S + t];
}

// BurgFeature::autoCorrelation (lpc.cc:158-207).  Scratch: EF, EB dim; A, Af order+1 (A is the output).
__global__ void k_lpc_burg(const float* __restrict__ XT, int Tc, int dim, int order, float* __restrict__ EF, float* __restrict__ EB,
                           float* __restrict__ A, float* __restrict__ Af, float* __restrict__ E0)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Tc) return;
  const size_t S = (size_t) Tc;
  float e0 = 0.0f;
  for (int j = 0; j < dim; j++) { const float x = XT[j * S + t]; e0 += x * x; EF[j * S + t] = x; EB[j * S + t] = x; }
  E0[t] = e0;
  for (int i = 0; i <= order; i++) { Af[i * S + t] = 0.0f; A[i * S + t] = 0.0f; }
  for (int i = 0; i < order; i++) {
    const int n = dim - i - 1;
    double num = 0.0, den = 0.0;
    for (int j = 0; j < n; j++) {
      const float efp = EF[(j + 1) * S + t], ebp = EB[j * S + t];
      num -= (double) (2 * ebp * efp);
      den += (double) (efp * efp + ebp * ebp);
    }
    const float k = (float) ((double) (float) num / den);
    for (int j = 0; j < n; j++) {
      const float efp = EF[(j + 1) * S + t], ebp = EB[j * S + t];
      EF[j * S + t] = efp + k * ebp; EB[j * S + t] = ebp + k * efp;
    }
    A[t] = 1.0f;
    for (int j = 0; j <= i + 1; j++) Af[j * S + t] = A[(i - j + 1) * S + t];
    for (int j = 1; j <= i + 1; j++) A[j * S + t] += k * Af[j * S + t];
  }
}

// MVDRFeature: the (order+1) distinct values of the symmetric sequence PC (lpc.h:157-170): V[i] = PC[order+i]
__global__ void k_lpc_mvdr_pc(const float* __restrict__ A, const float* __restrict__ E0, int Tc, int order, float* __restrict__ V)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Tc) return;
  const size_t S = (size_t) Tc; const bool pos = E0[t] > 0;
  for (int i = 0; i <= order; i++) {
    double temp = 0;
    for (int ii = 0; ii <= order - i; ii++) temp += (double) ((float) (order + 1 - i - 2 * ii) * A[ii * S + t] * A[(ii + i) * S + t]);
    V[i * S + t] = pos ? (float) -temp : 10000000.0f;
  }
}

// power spectrum of PA on the N-point grid + the envelope value.  kind 0: PA[n] = V[|n-1-order|], n = 1..2 order+1;
// kind 1: PA[n] = A[n-1], n = 1..order+1 (the shift by one "because of fft", lpc.h:172-174,318-319).
__global__ void k_lpc_envelope(const float* __restrict__ C, const float* __restrict__ E0, int Tc, int dim, int order, int N, int kind,
                               const double2* __restrict__ tw, double* __restrict__ out)
{
  const int outN = dim / 2 + 1;
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long) Tc * outN) return;
  const int t = (int) (idx / outN), k = (int) (idx - (long) t * outN);
  const size_t S = (size_t) Tc;
  const int L = (kind == 0) ? 2 * order + 1 : order + 1;
  double re = 0.0, im = 0.0;
  for (int n = 1; n <= L && n < dim; n++) {                    // (fftPower reads power[0..dim) only)
    const int ci = (kind == 0) ? ((n - 1 - order) < 0 ? order + 1 - n : n - 1 - order) : n - 1;
    const double v = (double) C[ci * S + t];
    const double2 w = tw[(int) (((long) n * k) % N)];
    re += v * w.x; im -= v * w.y;
  }
  const float p = (k == 0 || k == N / 2) ? (float) (re * re) : (float) (re * re + im * im);
  const float e0 = E0[t];
  double o;
  if (kind == 0) { o = sqrt((double) p); o = (o > 0) ? (double) e0 / o : 10000000.0; }
  else { o = (double) p; o = (o > 0) ? (double) (2 * e0) / (o * (double) dim) : 10000000.0; }
  out[idx] = o;
}

struct LpcPlan {
  int dim = 0, order = 0, method = 0, kind = 0, N = 0; float warp = 0.f;
  DevBuf<float> xt, s1, s2, r, a0, a1, lp, e0, v; DevBuf<double2> tw;
};

}  // namespace dsr

using namespace dsr;
struct dsr_lpc : LpcPlan {};

extern "C" {

dsr_status dsr_lpc_create(int dim, int order, int correlate, float warp, int method, int kind, dsr_lpc** out)
{
  return guard([&] {
    (void) correlate;                                          // stored but never used by the reference (lpc.h:124,302)
    if (!out) throw Error(DSR_E_PARAMETER, "null argument");
    if (dim < 2 || order < 1) throw Error(DSR_E_PARAMETER, "bad dimension %d / order %d", dim, order);
    if (order >= dim / 2 + 1) throw Error(DSR_E_PARAMETER, "Order (%d) and dimension (%d) do not match.", order, dim / 2 + 1);   // lpc.h:126-127
    if (method < 0 || method > 1 || kind < 0 || kind > 1) throw Error(DSR_E_PARAMETER, "method/kind must be 0 or 1");
    require_device();
    dsr_lpc* p = new dsr_lpc(); p->dim = dim; p->order = order; p->warp = warp; p->method = method; p->kind = kind;
    const unsigned l2 = (unsigned) ceil(log((double) dim) / log(2.0));                 // lpc.cc:32-35
    p->N = 1 << l2;
    std::vector<double2> tw((size_t) p->N);
    for (int m = 0; m < p->N; m++) { const double a = 2.0 * M_PI * (double) m / (double) p->N; tw[m].x = cos(a); tw[m].y = sin(a); }
    p->tw.upload(tw);
    *out = p;
  });
}
void dsr_lpc_destroy(dsr_lpc* p) { delete p; }
int dsr_lpc_size(const dsr_lpc* p) { return p ? p->dim / 2 + 1 : 0; }

dsr_status dsr_lpc_run(dsr_lpc* p, const float* frames_dev, int64_t T, double* out_dev, void* stream)
{
  return guard([&] {
    if (!p || !frames_dev || !out_dev) throw Error(DSR_E_PARAMETER, "null argument");
    hipStream_t st = (hipStream_t) stream;
    const int dim = p->dim, order = p->order, outN = dim / 2 + 1;
    const int chunk = 65536;
    for (int64_t t0 = 0; t0 < T; t0 += chunk) {
      const int Tc = (int) ((T - t0 < chunk) ? T - t0 : chunk);
      const size_t S = (size_t) Tc;
      p->xt.reserve(S * dim); p->s1.reserve(S * dim); p->e0.reserve(S); p->lp.reserve(S * (order + 1));
      hipLaunchKernelGGL(k_lpc_transpose, dim3(cdiv(dim, 32), cdiv(Tc, 32)), dim3(32, 8), 0, st, frames_dev + t0 * dim, Tc, dim, p->xt.p);
      const int nb = cdiv(Tc, 64);
      if (p->method == 0) {
        p->r.reserve(S * (order + 1)); p->a0.reserve(S * (order + 1)); p->a1.reserve(S * (order + 1));
        hipLaunchKernelGGL(k_lpc_warp, dim3(nb), dim3(64), 0, st, p->xt.p, Tc, dim, order, p->warp, p->s1.p, p->r.p, p->a0.p, p->a1.p, p->lp.p, p->e0.p);
      } else {
        p->s2.reserve(S * dim); p->a0.reserve(S * (order + 1));
        hipLaunchKernelGGL(k_lpc_burg, dim3(nb), dim3(64), 0, st, p->xt.p, Tc, dim, order, p->s1.p, p->s2.p, p->lp.p, p->a0.p, p->e0.p);
      }
      const float* coef = p->lp.p;
      if (p->kind == 0) {
        p->v.reserve(S * (order + 1));
        hipLaunchKernelGGL(k_lpc_mvdr_pc, dim3(nb), dim3(64), 0, st, p->lp.p, p->e0.p, Tc, order, p->v.p);
        coef = p->v.p;
      }
      const long n = (long) Tc * outN;
      hipLaunchKernelGGL(k_lpc_envelope, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, coef, p->e0.p, Tc, dim, order, p->N, p->kind,
                         p->tw.p, out_dev + t0 * outN);
      DSR_HIP(hipGetLastError());
    }
  });
}

}  // extern "C"
