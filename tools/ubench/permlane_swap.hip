// v_permlane32_swap semantics probe (gfx950): r = swap(v, v) must give r[0] = the LOW half's value
// in every lane and r[1] = the HIGH half's value in every lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* out) {
  const float x = in[threadIdx.x];
  const unsigned xb = __builtin_bit_cast(unsigned, x);
  // NOTE: __builtin_amdgcn_permlane32_swap followed by float math: hipcc 7.2 (ROCm 7.2.0) used
  // result[0] for BOTH results here (v_fma v2, v1, v1 ; v_fmac v2, v1, v1 and both stores from v1),
  // whether or not the two operands were the same SSA value -> inline asm with the two hazard
  // wait states inside the string
  unsigned sw[2] = {xb, xb};
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(sw[0]), "+v"(sw[1]));
  const float x0 = __builtin_bit_cast(float, sw[0]), x1 = __builtin_bit_cast(float, sw[1]);
  float a2 = 0.f;
  a2 = fmaf(x0, x0, a2);
  a2 = fmaf(x1, x1, a2);
  out[threadIdx.x] = x0;
  out[64 + threadIdx.x] = x1;
  out[128 + threadIdx.x] = a2;
}
int main() {
  float h[64], o[192], *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = (float)(i + 1);
  hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    const float lo = h[i & 31], hi = h[32 + (i & 31)];
    if (o[i] != lo || o[64 + i] != hi || o[128 + i] != lo * lo + hi * hi) ++bad;
  }
  printf("permlane32_swap(v,v): lane 5 -> x0=%g x1=%g, lane 37 -> x0=%g x1=%g, a2=%g; %s\n", o[5], o[69], o[37],
         o[101], o[133], bad ? "MISMATCH" : "ok");
  return bad != 0;
}
