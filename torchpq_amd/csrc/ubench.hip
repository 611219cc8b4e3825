// Measurement utility: a streaming-read microkernel.  bench.py uses it to measure the box's
// sustained HBM read rate (roofline fractions are reported against BOTH the 8 TB/s
// spec peak and this measured stream peak).  Not part of the search path.
#include "common.h"

namespace tpq {

// every lane streams 16-byte words, the wave a contiguous 1 KiB per instruction; UNROLL
// independent loads in flight per lane; the xor-reduction keeps the loads alive and the sink
// write never happens for real data (a 2^-32 event that would be harmless anyway)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4* __restrict__ src,
                                                          int64_t n16, uint32_t* __restrict__ sink) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  for (; i < n16; i += stride) {
    acc ^= __builtin_nontemporal_load(src + i);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) atomicAdd(sink, 1u);
}

// the k-means update's access pattern without its compute: one wave per block reads, per tile of
// 64 points, `rows` rows x 256 bytes (a lane 16 bytes, 16 lanes a row), rows `row_stride` floats
// apart; tiles dealt round-robin over gridDim.x blocks; two tiles of loads in flight
__global__ __launch_bounds__(64) void rows_read_kernel(const float* __restrict__ src, int rows,
                                                       int64_t row_stride, int64_t n,
                                                       uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x;
  const float* __restrict__ base = src + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * rows * row_stride;
  const int64_t step = (int64_t)gridDim.x * 64;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int64_t p0 = (int64_t)blockIdx.x * 64; p0 < n; p0 += 2 * step) {
    u32x4 v[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t pt = p0 + t * step + (lane & 15) * 4;
      const float* p = base + (pt < n ? pt : (int64_t)blockIdx.x * 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 4 * j + (lane >> 4);
        v[t][j] = *reinterpret_cast<const u32x4*>(p + (int64_t)(row < rows ? row : 0) * row_stride);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[t][j];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) atomicAdd(sink, 1u);
}

}  // namespace tpq

using namespace tpq;

extern "C" int tpq_ubench_rows_read(const float* src, int l, int d, int64_t n, int chunks,
                                    void* sink_or_null, tpq_stream_t stream) {
  TPQ_REQUIRE(src && l >= 1 && d >= 32 && d % 32 == 0 && n >= 64 && n % 4 == 0 && chunks >= 1,
              "ubench_rows_read: bad arguments");
  hipLaunchKernelGGL(rows_read_kernel, dim3((unsigned)chunks, (unsigned)(d / 32), (unsigned)l), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), src, 32, n, n,
                     reinterpret_cast<uint32_t*>(sink_or_null));
  TPQ_LAUNCH_CHECK("rows_read_kernel");
  return TPQ_OK;
}

extern "C" int tpq_ubench_stream_read(const void* src, size_t bytes, void* sink_or_null,
                                      int n_blocks, tpq_stream_t stream) {
  TPQ_REQUIRE(src, "ubench_stream_read: null pointer");
  TPQ_REQUIRE(bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0,
              "ubench_stream_read: buffer must be 16-byte aligned and a multiple of 16 bytes");
  TPQ_REQUIRE(n_blocks >= 0, "ubench_stream_read: n_blocks=%d", n_blocks);
  if (bytes == 0) return TPQ_OK;
  if (n_blocks == 0) n_blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 32 waves per CU
  hipLaunchKernelGGL(stream_read_kernel<8>, dim3((unsigned)n_blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const u32x4*>(src),
                     (int64_t)(bytes / 16), reinterpret_cast<uint32_t*>(sink_or_null));
  TPQ_LAUNCH_CHECK("stream_read_kernel");
  return TPQ_OK;
}
