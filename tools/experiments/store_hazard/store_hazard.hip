// Round-4 experiment: how soon after a 16-byte store may a VALU instruction overwrite the store's data registers?
// (standalone; built by build.sh here, run on the GPU box: tools/experiments/store_hazard/run.sh)
//
// A wave stores v[20:23] = (good, good, good, good) and, PAD instructions later, writes a poison word to v20 (the first
// data register) -- the sequence hipcc emitted in probe_sims_kernel's epilogue when it scheduled conversions between
// the stores.  Every poison word found in memory afterwards is a store that read its data after the overwrite.
//   KIND 0: buffer_store_dwordx4 ... offen, lanes in different rows (the sims matrix: lane = query, row stride = n_cells x 2 B)
//   KIND 1: buffer_store_dwordx4 ... offen, lanes contiguous (16 B apart)
//   KIND 2: global_store_dwordx4, lanes in different rows
//   KIND 3: KIND 0 with two MFMAs issued just before the store and v_max_f32 as the overwriting instruction
//   PAD n < 100: a chain of n DEPENDENT v_mul_f32 between the store and the overwrite;  PAD 100 + n: s_nop n;
//   PAD 200 + n: n INDEPENDENT v_mul_f32 / v_max_f32 (what the compiler's schedule had: two of them)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr uint32_t kGood = 0x3c003c00u, kPoison = 0x7f7fbeefu;  // (a finite float: v_max_f32 returns it unchanged)
constexpr int kIters = 32, kStoresPerIter = 4;

template <int PAD> struct PadStr;
#define PADSTR(n, s) template <> struct PadStr<n> { static constexpr const char* v() { return s; } };

template <int KIND, int PAD>
__global__ __launch_bounds__(256) void hazard_kernel(uint32_t* out, int row_bytes, int rows_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wave w of block b owns rows [b * rows_per_block + 64 w, +64) ; iteration it, store s: 16 bytes at column (it * 4 + s) * 16
  char* base = reinterpret_cast<char*>(out) + (int64_t)blockIdx.x * rows_per_block * row_bytes;   // the block's rows
  const int lane_off = KIND == 1 ? (wave * 64 + lane) * 16 : (wave * 64 + lane) * row_bytes;
  const int step = KIND == 1 ? 4096 : 16;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint64_t ba = reinterpret_cast<uint64_t>(base);
  const u32x4 rsrc = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ba),
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ba >> 32)),
                      (uint32_t)__builtin_amdgcn_readfirstlane(rows_per_block * row_bytes), 0x00020000u};
  uint32_t good = kGood, poison = kPoison;
  float junk = (float)lane;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int s = 0; s < kStoresPerIter; ++s) {
      const int voff = lane_off + (it * kStoresPerIter + s) * step;
      char* gp = base + voff;
#define BODY(PADTXT)                                                                                                  \
  if constexpr (KIND == 3)                                                                                            \
    asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\t"                 \
                 "v_mfma_f32_32x32x8_f16 a[0:15], v[28:29], v[30:31], a[0:15]\n\t"                                      \
                 "v_mfma_f32_32x32x8_f16 a[16:31], v[28:29], v[30:31], a[16:31]\n\ts_nop 4\n\t"                         \
                 "buffer_store_dwordx4 v[20:23], %0, %4, 0 offen\n\t" PADTXT "v_max_f32 v20, %2, %2\n\t"               \
                 : "+v"(*const_cast<int*>(&voff)), "+v"(good), "+v"(poison), "+v"(junk)                                \
                 : "s"(rsrc)                                                                                          \
                 : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "memory");  \
  else if constexpr (KIND == 2)                                                                                       \
    asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\ts_nop 4\n\t"       \
                 "global_store_dwordx4 %0, v[20:23], off\n\t" PADTXT "v_mov_b32 v20, %2\n\t"                           \
                 : "+v"(gp), "+v"(good), "+v"(poison), "+v"(junk)::"v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");      \
  else                                                                                                                \
    asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\ts_nop 4\n\t"       \
                 "buffer_store_dwordx4 v[20:23], %0, %4, 0 offen\n\t" PADTXT "v_mov_b32 v20, %2\n\t"                   \
                 : "+v"(*const_cast<int*>(&voff)), "+v"(good), "+v"(poison), "+v"(junk)                                \
                 : "s"(rsrc)                                                                                          \
                 : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
      if constexpr (PAD == 0) { BODY("") }
      else if constexpr (PAD == 1) { BODY("v_mul_f32 %3, %3, %3\n\t") }
      else if constexpr (PAD == 2) { BODY("v_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\t") }
      else if constexpr (PAD == 3) { BODY("v_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\tv_mul_f32 %3, %3, %3\n\t") }
      else if constexpr (PAD == 4) { BODY("v_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\tv_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\t") }
      else if constexpr (PAD == 6) { BODY("v_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\tv_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\tv_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\t") }
      else if constexpr (PAD == 8) { BODY("v_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\tv_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\tv_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\tv_mul_f32 %3, %3, %3\n\tv_mul_f32 v24, %3, %3\n\t") }
      else if constexpr (PAD == 202) { BODY("v_mul_f32 v24, %3, %3\n\tv_max_f32 v25, %3, %3\n\t") }
      else if constexpr (PAD == 203) { BODY("v_mul_f32 v24, %3, %3\n\tv_max_f32 v25, %3, %3\n\tv_mul_f32 v26, %3, %3\n\t") }
      else if constexpr (PAD == 204) { BODY("v_mul_f32 v24, %3, %3\n\tv_max_f32 v25, %3, %3\n\tv_mul_f32 v26, %3, %3\n\tv_max_f32 v27, %3, %3\n\t") }
      else if constexpr (PAD == 206) { BODY("v_mul_f32 v24, %3, %3\n\tv_max_f32 v25, %3, %3\n\tv_mul_f32 v26, %3, %3\n\tv_max_f32 v27, %3, %3\n\tv_mul_f32 v24, %3, %3\n\tv_max_f32 v25, %3, %3\n\t") }
      else if constexpr (PAD == 100) { BODY("s_nop 0\n\t") }
      else if constexpr (PAD == 101) { BODY("s_nop 1\n\t") }
      else if constexpr (PAD == 103) { BODY("s_nop 3\n\t") }
      else if constexpr (PAD == 107) { BODY("s_nop 7\n\t") }
    }
  }
  if (junk == 12345.678f) out[0] = 0;
}

__global__ void count_kernel(const uint32_t* p, size_t n, unsigned long long* cnt) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] == kPoison;
  if (c) atomicAdd(cnt, c);
}

template <int KIND, int PAD>
void run(uint32_t* buf, size_t bytes, int n_blocks, int row_bytes, unsigned long long* cnt) {
  CK(hipMemset(buf, 0, bytes));
  CK(hipMemset(cnt, 0, 8));
  hipLaunchKernelGGL((hazard_kernel<KIND, PAD>), dim3(n_blocks), dim3(256), 0, 0, buf, row_bytes, 256);
  CK(hipGetLastError());
  hipLaunchKernelGGL(count_kernel, dim3(2048), dim3(256), 0, 0, buf, bytes / 4, cnt);
  unsigned long long h = 0;
  CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
  const double stores = (double)n_blocks * 256 * kIters * kStoresPerIter;
  printf("{\"kind\": %d, \"pad\": %d, \"stores\": %.0f, \"poisoned_first_dwords\": %llu, \"fraction\": %.6f}\n", KIND, PAD, stores, h,
         (double)h / stores);
}

int main() {
  const int row_bytes = 8192, n_blocks = 1280;   // 1280 blocks x 256 rows x 8 KiB = 2.5 GiB ... keep 2 KiB used per row
  const size_t bytes = (size_t)n_blocks * 256 * row_bytes;
  uint32_t* buf;
  unsigned long long* cnt;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&cnt, 8));
#define RUNK(K) run<K, 0>(buf, bytes, n_blocks, row_bytes, cnt); run<K, 1>(buf, bytes, n_blocks, row_bytes, cnt); \
  run<K, 2>(buf, bytes, n_blocks, row_bytes, cnt); run<K, 3>(buf, bytes, n_blocks, row_bytes, cnt);                 \
  run<K, 4>(buf, bytes, n_blocks, row_bytes, cnt); run<K, 6>(buf, bytes, n_blocks, row_bytes, cnt);                 \
  run<K, 8>(buf, bytes, n_blocks, row_bytes, cnt); run<K, 100>(buf, bytes, n_blocks, row_bytes, cnt);               \
  run<K, 101>(buf, bytes, n_blocks, row_bytes, cnt); run<K, 103>(buf, bytes, n_blocks, row_bytes, cnt);             \
  run<K, 107>(buf, bytes, n_blocks, row_bytes, cnt); run<K, 202>(buf, bytes, n_blocks, row_bytes, cnt);             \
  run<K, 203>(buf, bytes, n_blocks, row_bytes, cnt); run<K, 204>(buf, bytes, n_blocks, row_bytes, cnt); run<K, 206>(buf, bytes, n_blocks, row_bytes, cnt);
  RUNK(0)
  RUNK(1)
  RUNK(2)
  RUNK(3)
  return 0;
}
