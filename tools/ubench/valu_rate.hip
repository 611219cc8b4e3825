// Issue rate of the VALU instructions of the top-2 update (v_and_or_b32, v_med3_f32, v_max3_f32, v_max_f32,
// v_min_f32) on gfx950: cycles per wave64 instruction, 8 independent chains per wave, 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(float* out, int iters, long long* cyc) {
  float a[8], b[8], c[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = threadIdx.x * 0.5f + i;
    b[i] = threadIdx.x * 0.25f - i;
    c[i] = 1.0f + i;
  }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define BODY(i)                                                                                              \
  if (OP == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                \
  if (OP == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));                 \
  if (OP == 2) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));                 \
  if (OP == 3) asm volatile("v_and_or_b32 %0, %0, -16, 3" : "+v"(a[i]));                                      \
  if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));                  \
  if (OP == 5) asm volatile("v_med3_f32 %2, %0, %1, %3\n v_max3_f32 %0, %0, %1, %3\n v_max_f32 %1, %1, %2" \
                            : "+v"(a[i]), "+v"(b[i]), "+v"(c[i]) : "v"(a[(i + 1) & 7]));
    REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY)
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i] + c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_body) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {
    // 256 CUs x 4 SIMDs x wps waves, one wave per block
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, 10, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 32 * instr_per_body;
    printf("%-28s waves/SIMD %d: %6.2f clock64 ticks per instr per wave, %.3f ms, %.2f ns per instr per SIMD\n", name, wps,
           (double)c / n, ms, ms * 1e6 / (n * wps));
  }
}

int main() {
  run<0>("v_max_f32", 1);
  run<1>("v_max3_f32", 1);
  run<2>("v_med3_f32", 1);
  run<3>("v_and_or_b32", 1);
  run<4>("v_fma_f32", 1);
  run<5>("med3+max3+max (top-2 pair)", 3);
  return 0;
}
