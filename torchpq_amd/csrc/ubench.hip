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

// The scan's own access pattern without its compute (the grid-stride kernel above read 6.45-6.5 TB/s
// while the 100 M-slot scan pulled 7.0-7.1 TB/s of fabric-side traffic): the buffer is cut into CHUNKS (a probed cell:
// `chunk16` 16-byte words); chunks are dealt round-robin to the workgroups, a chunk's 1-KiB pieces round-robin to the
// workgroup's waves, UNROLL pieces in flight per wave; NT picks non-temporal loads.
template <int UNROLL, bool NT>
__global__ __launch_bounds__(1024) void chunk_read_kernel(const u32x4* __restrict__ src, int64_t n16, int64_t chunk16,
                                                         uint32_t* __restrict__ sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int64_t n_chunks = (n16 + chunk16 - 1) / chunk16;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t base = c * chunk16, end = base + chunk16 < n16 ? base + chunk16 : n16;
    int64_t i = base + (int64_t)wave * 64 + lane;
    const int64_t step = (int64_t)nw * 64;
    for (; i + (UNROLL - 1) * step < end; i += UNROLL * step) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * step) : src[i + u * step];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    for (; i < end; i += step) acc ^= NT ? __builtin_nontemporal_load(src + i) : src[i];
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

// `tpq_ubench_stream_read` with the knobs exposed (bench.py sweeps a handful and quotes the best as the box's stream
// peak): threads per workgroup (64 ... 1024, multiple of 64), loads in flight per lane (4, 8, 16), chunk_bytes = 0 for
// the grid-stride walk or the size of a "cell" dealt to one workgroup, non-temporal loads or plain ones.
extern "C" int tpq_ubench_stream_read_ex(const void* src, size_t bytes, void* sink_or_null, int n_blocks,
                                         int threads, int unroll, size_t chunk_bytes, int nontemporal,
                                         tpq_stream_t stream) {
  TPQ_REQUIRE(src, "ubench_stream_read_ex: null pointer");
  TPQ_REQUIRE(bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && chunk_bytes % 16 == 0,
              "ubench_stream_read_ex: buffer and chunk must be 16-byte aligned multiples of 16 bytes");
  TPQ_REQUIRE(n_blocks >= 1 && threads >= 64 && threads <= 1024 && threads % 64 == 0,
              "ubench_stream_read_ex: n_blocks=%d threads=%d", n_blocks, threads);
  TPQ_REQUIRE(unroll == 4 || unroll == 8 || unroll == 16, "ubench_stream_read_ex: unroll=%d (4, 8, 16)", unroll);
  if (bytes == 0) return TPQ_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const u32x4* p = reinterpret_cast<const u32x4*>(src);
  uint32_t* sink = reinterpret_cast<uint32_t*>(sink_or_null);
  const int64_t n16 = (int64_t)(bytes / 16), c16 = chunk_bytes ? (int64_t)(chunk_bytes / 16) : n16;
  // (chunk_bytes = 0: ONE chunk would serialise on one workgroup -- the walk below deals 1-KiB pieces to all waves of
  // the grid instead: a chunk per workgroup of bytes / n_blocks, rounded up to whole KiB)
  const int64_t per = chunk_bytes ? c16 : (((n16 + n_blocks - 1) / n_blocks + 63) / 64) * 64;
#define TPQ_UB(U, NT) hipLaunchKernelGGL((chunk_read_kernel<U, NT>), dim3((unsigned)n_blocks), dim3((unsigned)threads), \
                                         0, st, p, n16, per, sink)
  if (nontemporal) {
    if (unroll == 4) TPQ_UB(4, true); else if (unroll == 8) TPQ_UB(8, true); else TPQ_UB(16, true);
  } else {
    if (unroll == 4) TPQ_UB(4, false); else if (unroll == 8) TPQ_UB(8, false); else TPQ_UB(16, false);
  }
#undef TPQ_UB
  TPQ_LAUNCH_CHECK("chunk_read_kernel");
  return TPQ_OK;
}
