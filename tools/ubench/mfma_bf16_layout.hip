// Which lane / element of the A and B operands of v_mfma_f32_32x32x16_bf16 carries A[row][k] and
// B[k][col]?  Hypothesis (the K=16 analogue of the 32x32x2 f32 form used all over csrc/):
//   A: lane = row + 32 * (k / 8), element k % 8;   B: lane = col + 32 * (k / 8), element k % 8;
//   D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// One wave per (row, k): A is one-hot at (row, k), B[k][col] = k + 1 (exact in bf16), then
// B[k][col] = col + 1; D[row][col] must be k + 1 resp. col + 1 and zero elsewhere.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_bf16_layout.hip -o tools/ubench/mfma_bf16_layout
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(int* bad) {
  const int r0 = blockIdx.x / 16, k0 = blockIdx.x % 16;
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  for (int mode = 0; mode < 2; ++mode) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * half + i;
      a[i] = (__bf16)((l31 == r0 && k == k0) ? 1.0f : 0.0f);
      b[i] = (__bf16)(mode == 0 ? (float)(k + 1) : (float)(l31 + 1));
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half, col = l31;
      const float want = row == r0 ? (mode == 0 ? (float)(k0 + 1) : (float)(col + 1)) : 0.f;
      if (c[r] != want) atomicAdd(bad, 1);
    }
  }
}

int main() {
  int* d_bad;
  int h_bad = -1;
  hipMalloc(&d_bad, 4);
  hipMemset(d_bad, 0, 4);
  hipLaunchKernelGGL(probe, dim3(32 * 16), dim3(64), 0, 0, d_bad);
  hipMemcpy(&h_bad, d_bad, 4, hipMemcpyDeviceToHost);
  printf("mismatches: %d (0 = the hypothesised layout holds)\n", h_bad);
  return h_bad != 0;
}
