// csrc/ops.h -- launchers of the per-operator kernels (k_ops.hip) and table builders shared with k_mfcc.hip.
#pragma once
#include "common.h"
namespace dsr {
void op_frames(const float* x, int nsamp, int T, int L, int shift, float* out, hipStream_t st);
void op_preemph(const float* in, int T, int L, double mu, float* out, hipStream_t st);
void op_hamming_f(const float* in, int T, int L, const double* w, float* out, hipStream_t st);
void op_hamming_s(const short* in, int T, int L, const double* w, float* out, hipStream_t st);
void op_fft(const float* in, int T, int L, int fftLen, const double2* tw, double2* out, hipStream_t st);
void op_power(const double2* in, int T, int fftLen, int powN, double* out, hipStream_t st);
void op_vtln(const double* in, int T, int N, const int* s, const int* c, const int* o, const double* coef, const double* div, int rf, double* out, hipStream_t st);
void op_mel(const double* in, int T, int N, int filterN, const int* s, const int* c, const int* o, const float* coef, double* out, hipStream_t st);
void op_log(const double* in, long n, double m, double a, int sphinx, float* out, hipStream_t st);
void op_sgemv(const float* in, int T, int cols, int rows, const float* A, float* out, hipStream_t st);
void op_adjacent(const float* in, int T, int N, int delta, float* out, hipStream_t st);
void op_expand_bins(const float2* in, int T, int F, int M, double2* out, hipStream_t st);
void op_pack_bins(const double2* in, int T, int F, int M, float2* out, hipStream_t st);
void op_pack_hermitian(const double2* in, int T, int M, float2* out, hipStream_t st);
void op_highpass(const double2* in, int T, int M, int cutBin, double2* out, hipStream_t st);
void op_orth_assemble(const float2* low, const double2* full, int T, int F, int M, double2* out, hipStream_t st);
void op_link_ac(const float* scores, int K, const int* dist, const int* start, const int* end, int n, double* out, hipStream_t st);
void op_cmn(const float* in, int T, int N, int mode, double devNormFactor, float* out, hipStream_t st, const float* wgt = nullptr, int wStride = 0);   // k_mfcc.hip
// host table builders (k_mfcc.hip), reference formulas of feature.cc:1726-1838,1954-2090 and gslmatrix.cc:108-132
struct SparseRowsD { std::vector<int> start, count, off; std::vector<double> coef, div; int roundFloat = 0; };
struct SparseRowsF { std::vector<int> start, count, off; std::vector<float> coef; int nReq = 0; };
void build_vtln_rows(int N, double ratio, double edge, int version, SparseRowsD& r);
void build_mel_rows(int powN, float rate, float low, float up, int filterN, int version, SparseRowsF& r);
void build_dct(int ncep, int nmel, int type, std::vector<float>& m);
}
