// Probe: what one "phase" of a persistent 1024-thread workgroup costs on gfx950 -- the fixed price of exchanging data between the threads of
// a workgroup (a) through global memory (store, barrier, load what a neighbour wrote), (b) through LDS, (c) a bare barrier, (d) one dependent
// global gather from a 6 MB table.  One workgroup per CU, 256 CUs, time per iteration from wall_clock64 (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(int mode, int iters, float4* buf, const float4* table, int tableN, long long* out)
{
  __shared__ float4 lds[1024];
  const int tid = threadIdx.x; float4* mine = buf + (size_t) blockIdx.x * 2048;
  float4 v = make_float4(tid, 1.f, 2.f, 3.f); unsigned idx = tid * 2654435761u;
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    if (mode == 0) { mine[(it & 1) * 1024 + tid] = v; __syncthreads(); v = mine[(it & 1) * 1024 + ((tid + 577) & 1023)]; v.x += 1.f; __syncthreads(); }
    else if (mode == 1) { lds[tid] = v; __syncthreads(); v = lds[(tid + 577) & 1023]; v.x += 1.f; __syncthreads(); }
    else if (mode == 2) { v.x += 1.f; __syncthreads(); __syncthreads(); }
    else if (mode == 3) { idx = idx * 1664525u + 1013904223u; const float4 t = table[idx % (unsigned) tableN]; v.x += t.x; idx += (unsigned) t.y; __syncthreads(); __syncthreads(); }
    else if (mode == 4) { mine[(it & 1) * 1024 + tid] = v; __syncthreads(); v.x += 1.f; __syncthreads(); }                                   // store + barrier, no read back
    else if (mode == 5) { v = mine[(it & 1) * 1024 + ((tid + 577) & 1023)]; v.x += 1.f; __syncthreads(); __syncthreads(); }                 // load (L2 hit) + barrier
  }
  const long long t1 = wall_clock64();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (v.x == -1.f) buf[0] = v;
}
int main()
{
  const int B = 256, iters = 2000, tableN = 6 * 1024 * 1024 / 16;
  float4 *buf, *table; long long* out;
  hipMalloc(&buf, sizeof(float4) * 2048 * B); hipMalloc(&table, sizeof(float4) * tableN); hipMalloc(&out, 8 * B);
  hipMemset(buf, 0, sizeof(float4) * 2048 * B); hipMemset(table, 0, sizeof(float4) * tableN);
  const char* names[] = {"global store -> barrier -> global load of a neighbour's value -> barrier", "LDS store -> barrier -> LDS load -> barrier", "two bare barriers",
                         "dependent gather from a 6 MB table + two barriers", "global store -> two barriers", "global load (written long ago) -> two barriers"};
  for (int mode = 0; mode < 6; mode++) {
    hipLaunchKernelGGL(k, dim3(B), dim3(1024), 0, 0, mode, iters, buf, table, tableN, out); hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(B), dim3(1024), 0, 0, mode, iters, buf, table, tableN, out); hipDeviceSynchronize();
    std::vector<long long> h(B); hipMemcpy(h.data(), out, 8 * B, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double) x;
    printf("mode %d: %.3f us per iteration  (%s)\n", mode, s / B / iters / 100.0, names[mode]);
  }
  return 0;
}
