// LDS float/int atomic throughput probe (gfx950): which access patterns are cheap?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void probe(const int* __restrict__ idx, float* out, int iters) {
  __shared__ float bins[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) bins[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int a = idx[blockIdx.x * 256 + threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    int adr;
    if (MODE == 0) adr = (lane + it * 64) & 4095;              // consecutive, unique
    else if (MODE == 1) adr = (a + it * 977) & 255;            // random among 256 bins
    else if (MODE == 2) adr = ((a + it * 977) & 255) * 16 + (lane & 15);  // 4 labels x 16 dims
    else if (MODE == 3) adr = (a + it * 977) & 4095;           // random among 4096
    else adr = (a + it * 977) & 255;                           // int atomics, 256 bins
    if (MODE == 4) atomicAdd(reinterpret_cast<int*>(&bins[adr]), 1);
    else atomicAdd(&bins[adr], 1.0f);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = bins[0];
}

template <int MODE>
void run(const char* name, const int* idx, float* out) {
  const int blocks = 256 * 8, iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, idx, out, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, idx, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double lane_ops = (double)blocks * 256 * iters;
  printf("%-34s %8.3f ms  %8.2f G lane-atomics/s  (%.2f per clk per CU @2.1GHz)\n", name, ms,
         lane_ops / ms / 1e6, lane_ops / ms / 1e6 / 256 / 2.1);
}

int main() {
  int* idx;
  float* out;
  const int n = 256 * 8 * 256;
  hipMalloc(&idx, n * 4);
  hipMalloc(&out, 4096 * 4);
  int* h = (int*)malloc(n * 4);
  uint32_t s = 12345;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) & 0xffff; }
  hipMemcpy(idx, h, n * 4, hipMemcpyHostToDevice);
  run<0>("f32 consecutive unique", idx, out);
  run<1>("f32 random 256 bins", idx, out);
  run<2>("f32 4 labels x 16 dims", idx, out);
  run<3>("f32 random 4096 bins", idx, out);
  run<4>("i32 random 256 bins", idx, out);
  return 0;
}
