// Issue rates of the decoder's arithmetic on gfx950: fp32 add, fp64 add, fp64 mul, f32 -> f64 and f64 -> f32 conversions, v_mul_lo_u32, v_readlane.
// Each thread runs 8 independent chains of `iters` operations; 1024 threads per block, one block per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ __launch_bounds__(1024) void k(float* out, int iters, float seed)
{
  float a[8]; double d[8]; unsigned u[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; d[i] = a[i]; u[i] = (unsigned) a[i]; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = a[i] + 1.25f;
      if (OP == 1) d[i] = d[i] + 1.25;
      if (OP == 2) d[i] = d[i] * 1.0000001;
      if (OP == 3) { d[i] = (double) a[i]; a[i] = a[i] + (float) (int) (d[i] > 3.0); }       // cvt f32->f64 (+ cmp, cvt int, add)
      if (OP == 4) { a[i] = (float) d[i]; d[i] = d[i] + (double) (a[i] > 3.0f); }             // cvt f64->f32 (+ cmp, cvt, add f64)
      if (OP == 5) u[i] = u[i] * 2654435761u + 1u;
      if (OP == 6) a[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[i]), 5)) + 1.0f;
    }
  }
  float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float) d[i] + (float) u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
  float* out; hipMalloc(&out, 256 * 1024 * 4); const int iters = 4096;
  const char* nm[7] = {"f32 add", "f64 add", "f64 mul", "cvt f32->f64 (+3 ops)", "cvt f64->f32 (+3 ops)", "u32 mul_lo + add", "readlane + add"};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(OP) { hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, 16, 1.0f); hipDeviceSynchronize(); hipEventRecord(e0); \
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
  printf("%-26s %.3f ms  -> %.2f cycles per wave-op group at 2.4 GHz (per SIMD: 4 waves)\n", nm[OP], ms, ms * 1e-3 * 2.4e9 / (iters * 8.0 * 4.0)); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
  return 0;
}
