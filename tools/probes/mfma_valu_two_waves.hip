// probe: do the VALU instructions of ONE wave overlap the MFMAs of ANOTHER wave on the same SIMD?  512 threads a workgroup (two waves per SIMD),
// waves 0-3 run MFMAs only, waves 4-7 VALU only; timed: each kind alone, then both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV>
__global__ __launch_bounds__(512, 1) void k(float* out, int itersM, int itersV, float seed)
{
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  float s = 0;
  if (wave < 4) {
    f32x16 acc[4]; for (int t = 0; t < 4; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
    float a = seed + threadIdx.x, b = seed * 2.0f;
    for (int it = 0; it < itersM; it++) {
#pragma unroll
      for (int t = 0; t < 4; t++) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
    for (int t = 0; t < 4; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
  } else {
    float v[8]; for (int i = 0; i < 8; i++) v[i] = seed + i;
    const float b = seed * 2.0f;
    for (int it = 0; it < itersV; it++) {
#pragma unroll
      for (int n = 0; n < 4 * NV; n++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[n & 7]) : "v"(b));
    }
    for (int i = 0; i < 8; i++) s += v[i];
  }
  if (s == 123.456f) out[threadIdx.x] = s + lds[threadIdx.x];
}
template <int NV> void run(float* d)
{
  (void) hipFuncSetAttribute((const void*) k<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int iters = 20000; hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  const int cfg[3][2] = {{iters, 0}, {0, iters}, {iters, iters}}; float ms[3];
  for (int c = 0; c < 3; c++) {
    hipLaunchKernelGGL((k<NV>), dim3(256), dim3(512), 100 * 1024, 0, d, 100, 100, 1.0f);
    (void) hipEventRecord(e0); hipLaunchKernelGGL((k<NV>), dim3(256), dim3(512), 100 * 1024, 0, d, cfg[c][0], cfg[c][1], 1.0f); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    (void) hipEventElapsedTime(&ms[c], e0, e1);
  }
  printf("n=%2d VALU per MFMA: MFMA wave alone %.3f ms, VALU wave alone %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n", NV, ms[0], ms[1], ms[2], ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1]);
}
int main()
{
  float* d; (void) hipMalloc(&d, 4096);
  run<4>(d); run<8>(d); run<12>(d); run<16>(d); run<24>(d);
  return 0;
}
