// csrc/k_ops.hip -- one device kernel per feature operator, used by the stream/feature-operator API
// (dsr_stream_*, include/dsr.h section 7).  The batch pipe uses the fused kernels of k_mfcc.hip; these serve
// arbitrary operator chains built through the reference's own interface (btk/feature/feature.h), one
// utterance at a time, with every operator's output type and precision as in the reference.
// Compiled with -ffp-contract=off.
#include "common.h"
#include "ops.h"
#include <cmath>

namespace dsr {

// SampleFeature::next (feature.cc:610-659): rows of blockLen samples every shiftLen, zero padded
__global__ void k_op_frames(const float* x, int nsamp, int T, int L, int shift, float* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * L) return;
  const int t = (int) (idx / L), i = (int) (idx - (long) t * L); const long n = (long) t * shift + i;
  out[idx] = (n < nsamp) ? x[n] : 0.0f;
}
// PreemphasisFeature::next (feature.cc:1154-1170): prior carried across rows
__global__ void k_op_preemph(const float* in, int T, int L, double mu, float* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * L) return;
  const float b = in[idx]; const float prior = (idx == 0) ? 0.0f : in[idx - 1];     // rows are contiguous: [t][i-1] or [t-1][L-1]
  out[idx] = (float) __dsub_rn((double) b, __dmul_rn(mu, (double) prior));
}
// HammingFeature / HammingFeatureShort (feature.cc:1175-1232)
__global__ void k_op_hamming_f(const float* in, int T, int L, const double* w, float* out)
{ const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * L) return; out[idx] = (float) __dmul_rn(w[idx % L], (double) in[idx]); }
__global__ void k_op_hamming_s(const short* in, int T, int L, const double* w, float* out)
{ const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * L) return; out[idx] = (float) __dmul_rn(w[idx % L], (double) in[idx]); }

// FFTFeature::next (feature.cc:1266-1293): zero padded real FFT, unpacked to fftLen complex doubles with
// the conjugate mirror (halfComplexUnpack :46-60).  One workgroup of 64 threads per row, Stockham radix-2 in LDS.
__global__ __launch_bounds__(64) void k_op_fft(const float* in, int T, int L, int fftLen, const double2* tw, double2* out)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* a = reinterpret_cast<double2*>(smem); double2* b = a + fftLen;
  const int t = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < fftLen; i += 64) a[i] = make_double2(i < L ? (double) in[(long) t * L + i] : 0.0, 0.0);
  __syncthreads();
  double2* x = a; double2* y = b;
  for (int n = fftLen, s = 1; n > 1; n >>= 1, s <<= 1) {            // forward sign
    const int m = n >> 1;
    for (int j = lane; j < fftLen / 2; j += 64) {
      const int p = j / s, q = j - p * s;
      double2 w = tw[p * (fftLen / n)]; w.y = -w.y;
      const double2 u = x[q + s * p], v = x[q + s * (p + m)];
      y[q + s * (2 * p)] = make_double2(u.x + v.x, u.y + v.y);
      const double2 d = make_double2(u.x - v.x, u.y - v.y);
      y[q + s * (2 * p + 1)] = make_double2(d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x);
    }
    __syncthreads();
    double2* tmp = x; x = y; y = tmp;
  }
  const int len2 = fftLen / 2;
  for (int k = lane; k <= len2; k += 64) {
    double2 v = x[k];
    if (k == 0 || k == len2) v.y = 0.0;
    out[(long) t * fftLen + k] = v;
    if (k > 0 && k < len2) out[(long) t * fftLen + (fftLen - k)] = make_double2(v.x, -v.y);
  }
}
// SpectralPowerFeature::next (feature.cc:1329-1355)
__global__ void k_op_power(const double2* in, int T, int fftLen, int powN, double* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * powN) return;
  const int t = (int) (idx / powN), i = (int) (idx - (long) t * powN); const double2 z = in[(long) t * fftLen + i];
  out[idx] = __dadd_rn(__dmul_rn(z.x, z.x), __dmul_rn(z.y, z.y));
}
// VTLNFeature::next (feature.cc:1716-1838) as sparse interval weights
__global__ void k_op_vtln(const double* in, int T, int N, const int* start, const int* count, const int* off, const double* coef,
                          const double* div, int roundFloat, double* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * N) return;
  const int t = (int) (idx / N), k = (int) (idx - (long) t * N); const double* p = in + (long) t * N;
  double z = 0.0;
  for (int i = 0; i < count[k]; i++) { double v = p[start[k] + i]; if (roundFloat) v = (double) (float) v; z = __dadd_rn(z, __dmul_rn(coef[off[k] + i], v)); }
  if (div[k] != 0.0) z = z / div[k];
  out[idx] = z;
}
// MelFeature::next (feature.cc:2098-2160)
__global__ void k_op_mel(const double* in, int T, int N, int filterN, const int* start, const int* count, const int* off, const float* coef, double* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * filterN) return;
  const int t = (int) (idx / filterN), j = (int) (idx - (long) t * filterN); const double* a = in + (long) t * N + start[j]; const float* b = coef + off[j];
  const int n = count[j]; double sum = 0.0; int i = 0;
  for (; i + 4 <= n; i += 4) {
    double g = __dmul_rn(a[i], (double) b[i]); g = __dadd_rn(g, __dmul_rn(a[i + 1], (double) b[i + 1]));
    g = __dadd_rn(g, __dmul_rn(a[i + 2], (double) b[i + 2])); g = __dadd_rn(g, __dmul_rn(a[i + 3], (double) b[i + 3])); sum = __dadd_rn(sum, g);
  }
  for (; i < n; i++) sum = __dadd_rn(sum, __dmul_rn(a[i], (double) b[i]));
  out[idx] = sum;
}
// LogFeature::next (feature.cc:2398-2434)
__global__ void k_op_log(const double* in, long n, double m, double a, int sphinx, float* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= n) return;
  double v = in[idx];
  if (sphinx) { if (v < 1.0E-05) v = 1.0E-05; } else { v = __dadd_rn(v, a); if (v <= 0.0) v = 1.0; }
  out[idx] = (float) __dmul_rn(m, log10(v));
}
// gsl_blas_sgemv(NoTrans) reference loop: CepstralFeature (feature.cc:2479-2490), LinearTransformFeature (:2943-2957)
__global__ void k_op_sgemv(const float* in, int T, int cols, int rows, const float* A, float* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * rows) return;
  const int t = (int) (idx / rows), i = (int) (idx - (long) t * rows); const float* x = in + (long) t * cols; const float* a = A + (long) i * cols;
  float temp = 0.0f; for (int j = 0; j < cols; j++) temp = __fadd_rn(temp, __fmul_rn(x[j], a[j]));
  out[idx] = __fadd_rn(0.0f, temp);
}
// AdjacentFeature (feature.cc:2850-2904)
__global__ void k_op_adjacent(const float* in, int T, int N, int delta, float* out)
{
  const int W = (2 * delta + 1) * N; const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * W) return;
  const int t = (int) (idx / W), r = (int) (idx - (long) t * W), s = r / N, k = r - s * N;
  int src = t + s - delta; if (src < 0) src = 0; if (src > T - 1) src = T - 1;
  out[idx] = in[(long) src * N + k];
}
// complex64 [T][F] (bins 0..M/2) -> complex128 [T][M] with conjugate mirror (the stream's gsl_vector_complex of size M)
__global__ void k_op_expand_bins(const float2* in, int T, int F, int M, double2* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * M) return;
  const int t = (int) (idx / M), f = (int) (idx - (long) t * M);
  if (f < F) { const float2 v = in[(long) t * F + f]; out[idx] = make_double2(v.x, v.y); }
  else { const float2 v = in[(long) t * F + (M - f)]; out[idx] = make_double2(v.x, -v.y); }
}
// highPassFilter::next (postfilter.cc:1232-1256): bin 0 and bins 1..cut-1 are zero, bins cut..M/2 pass and are mirrored; the mirror bins of
// the cut ones stay at the zero the stream's vector was allocated with
__global__ void k_op_highpass(const double2* in, int T, int M, int cut, double2* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * M) return;
  const int t = (int) (idx / M), f = (int) (idx - (long) t * M), M2 = M / 2;
  double2 v = make_double2(0.0, 0.0);
  if (f <= M2) { if (f >= cut) v = in[(long) t * M + f]; }
  else { const int k = M - f; if (k >= cut) { const double2 q = in[(long) t * M + k]; v = make_double2(q.x, -q.y); } }
  out[idx] = v;
}
// SubbandOrthogonalizer::next with outChanX > 0 (beamformer.cc:2831-2849): blockingMatrixOutput writes bins 0..M/2 into the beamformer's own
// vector, whose upper bins still hold the mirror of the beamformer's output -- the operator hands out that vector
__global__ void k_op_orth_assemble(const float2* low, const double2* full, int T, int F, int M, double2* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * M) return;
  const int t = (int) (idx / M), f = (int) (idx - (long) t * M);
  if (f < F) { const float2 v = low[(long) t * F + f]; out[idx] = make_double2(v.x, v.y); } else out[idx] = full[idx];
}
// complex128 [T][M] -> complex64 [T][F]
__global__ void k_op_pack_bins(const double2* in, int T, int F, int M, float2* out)
{
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * F) return;
  const int t = (int) (idx / F), f = (int) (idx - (long) t * F); const double2 v = in[(long) t * M + f]; out[idx] = make_float2((float) v.x, (float) v.y);
}
// complex128 [T][M] -> complex64 [T][M/2+1] for the synthesis bank, which keeps the real part of the transform of ALL M bins (modulated.cc:598-610):
// that is the transform of the frame's Hermitian part (in[k] + conj(in[M-k])) / 2.  For a conjugate-symmetric frame (every analysis bank, every
// beamformer but SubbandMMI with the APAB post-filter) the sum is exact and the value is in[k] itself.
__global__ void k_op_pack_hermitian(const double2* in, int T, int M, float2* out)
{
  const int F = M / 2 + 1;
  const long idx = (long) blockIdx.x * blockDim.x + threadIdx.x; if (idx >= (long) T * F) return;
  const int t = (int) (idx / F), f = (int) (idx - (long) t * F); const double2 a = in[(long) t * M + f];
  if (f == 0 || 2 * f == M) { out[idx] = make_float2((float) a.x, (float) a.y); return; }
  const double2 b = in[(long) t * M + (M - f)];
  out[idx] = make_float2((float) ((a.x + b.x) * 0.5), (float) ((a.y - b.y) * 0.5));
}

#define GRID(n) dim3((unsigned) (((n) + 255) / 256)), dim3(256), 0, st
// Lattice::_updateAcNode (asr/lattice/lattice.cc:392-409): a link's acoustic score = the sum over its frames of its distribution's score, a double
// accumulator over float scores in frame order.  One thread per link walking a column of the score matrix [T][K]: a gather, a few KB per lattice.
__global__ void k_op_link_ac(const float* __restrict__ scores, int K, const int* __restrict__ dist, const int* __restrict__ start,
                             const int* __restrict__ end, int n, double* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double sum = 0.0;
  const float* col = scores + dist[i];
  for (int f = start[i]; f <= end[i]; f++) sum = __dadd_rn(sum, (double) col[(size_t) f * K]);
  out[i] = sum;
}

void op_frames(const float* x, int nsamp, int T, int L, int shift, float* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_frames, GRID((long) T * L), x, nsamp, T, L, shift, out); }
void op_preemph(const float* in, int T, int L, double mu, float* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_preemph, GRID((long) T * L), in, T, L, mu, out); }
void op_hamming_f(const float* in, int T, int L, const double* w, float* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_hamming_f, GRID((long) T * L), in, T, L, w, out); }
void op_hamming_s(const short* in, int T, int L, const double* w, float* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_hamming_s, GRID((long) T * L), in, T, L, w, out); }
void op_fft(const float* in, int T, int L, int fftLen, const double2* tw, double2* out, hipStream_t st)
{
  if (T <= 0) return;
  const size_t lds = 2 * (size_t) fftLen * sizeof(double2);
  DSR_HIP(hipFuncSetAttribute((const void*) k_op_fft, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  hipLaunchKernelGGL(k_op_fft, dim3(T), dim3(64), lds, st, in, T, L, fftLen, tw, out);
}
void op_power(const double2* in, int T, int fftLen, int powN, double* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_power, GRID((long) T * powN), in, T, fftLen, powN, out); }
void op_vtln(const double* in, int T, int N, const int* s, const int* c, const int* o, const double* coef, const double* div, int rf, double* out, hipStream_t st)
{ if (T > 0) hipLaunchKernelGGL(k_op_vtln, GRID((long) T * N), in, T, N, s, c, o, coef, div, rf, out); }
void op_mel(const double* in, int T, int N, int filterN, const int* s, const int* c, const int* o, const float* coef, double* out, hipStream_t st)
{ if (T > 0) hipLaunchKernelGGL(k_op_mel, GRID((long) T * filterN), in, T, N, filterN, s, c, o, coef, out); }
void op_log(const double* in, long n, double m, double a, int sphinx, float* out, hipStream_t st) { if (n > 0) hipLaunchKernelGGL(k_op_log, GRID(n), in, n, m, a, sphinx, out); }
void op_sgemv(const float* in, int T, int cols, int rows, const float* A, float* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_sgemv, GRID((long) T * rows), in, T, cols, rows, A, out); }
void op_adjacent(const float* in, int T, int N, int delta, float* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_adjacent, GRID((long) T * (2 * delta + 1) * N), in, T, N, delta, out); }
void op_expand_bins(const float2* in, int T, int F, int M, double2* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_expand_bins, GRID((long) T * M), in, T, F, M, out); }
void op_highpass(const double2* in, int T, int M, int cutBin, double2* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_highpass, GRID((long) T * M), in, T, M, cutBin, out); }
void op_orth_assemble(const float2* low, const double2* full, int T, int F, int M, double2* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_orth_assemble, GRID((long) T * M), low, full, T, F, M, out); }
void op_link_ac(const float* scores, int K, const int* dist, const int* start, const int* end, int n, double* out, hipStream_t st)
{ if (n > 0) hipLaunchKernelGGL(k_op_link_ac, GRID((long) n), scores, K, dist, start, end, n, out); }
void op_pack_hermitian(const double2* in, int T, int M, float2* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_pack_hermitian, GRID((long) T * (M / 2 + 1)), in, T, M, out); }
void op_pack_bins(const double2* in, int T, int F, int M, float2* out, hipStream_t st) { if (T > 0) hipLaunchKernelGGL(k_op_pack_bins, GRID((long) T * F), in, T, F, M, out); }
#undef GRID

}  // namespace dsr
