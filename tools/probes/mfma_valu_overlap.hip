// probe: can ONE wave per SIMD overlap its own VALU work with its own MFMAs?  A loop of (1 MFMA 32x32x2 f32 + n VALU) x 4 independent tiles,
// one 256-thread workgroup per CU (100 KB of LDS asked for, so no second workgroup fits), timed by the wall clock.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_probe tools/probes/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float seed)
{
  extern __shared__ float lds[];
  f32x16 acc[4]; for (int t = 0; t < 4; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
  float a = seed + threadIdx.x, b = seed * 2.0f;
  float v[8]; for (int i = 0; i < 8; i++) v[i] = seed + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
#pragma unroll
      for (int n = 0; n < NV; n++) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[n & 7]) : "v"(b));
        else if (KIND == 1) asm volatile("s_nop 0");
        else if (KIND == 2) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[n & 7]) : "v"(b) : "vcc");
        else if (KIND == 3) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
      }
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
  float s = 0; for (int t = 0; t < 4; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
  for (int i = 0; i < 8; i++) s += v[i];
  if (s == 123.456f) out[threadIdx.x] = s + lds[threadIdx.x];
}
template <int NV, int KIND> void run(float* d, const char* name)
{
  hipFuncSetAttribute((const void*) k<NV, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int iters = 20000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 100 * 1024, 0, d, 100, 1.0f);
  hipEventRecord(e0); hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 100 * 1024, 0, d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-10s n=%2d per MFMA: %.3f ms -> %.1f ns per (MFMA + n) = %.1f cycles at 2.4 GHz\n", name, (KIND == 2 ? 2 : 1) * NV, ms, ms * 1e6 / (iters * 4.0), ms * 1e6 / (iters * 4.0) * 2.4);
}
int main()
{
  float* d; hipMalloc(&d, 4096);
  run<0, 0>(d, "none"); run<4, 0>(d, "v_add"); run<8, 0>(d, "v_add"); run<12, 0>(d, "v_add"); run<14, 0>(d, "v_add"); run<16, 0>(d, "v_add"); run<20, 0>(d, "v_add"); run<24, 0>(d, "v_add"); run<32, 0>(d, "v_add");
  run<8, 1>(d, "s_nop"); run<16, 1>(d, "s_nop"); run<24, 1>(d, "s_nop");
  run<4, 2>(d, "cmp+cnd"); run<6, 2>(d, "cmp+cnd"); run<8, 2>(d, "cmp+cnd"); run<12, 2>(d, "cmp+cnd");
  run<8, 3>(d, "s_add"); run<16, 3>(d, "s_add"); run<24, 3>(d, "s_add");
  return 0;
}
