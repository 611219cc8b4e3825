// Reference code layout <-> MI355X scan layout (scan_layout.h), and the add-path scatter.
#include "common.h"
#include "scan_layout.h"

namespace tpq {

// one thread per (slot, output dword)
__global__ __launch_bounds__(256) void pack_codes_kernel(const uint8_t* __restrict__ codes,
                                                         uint8_t* __restrict__ packed,
                                                         int64_t n_slots, int m, int64_t slot_begin,
                                                         int64_t count) {
  const int G = m >> 2;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= count * G) return;
  const int d = (int)(t / count);  // output dword index inside the slot
  const int64_t s = slot_begin + (t - (int64_t)d * count);
  uint32_t w = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = d * 4 + u;
    const int j = scan_layout::subq_at(m, p, s);
    const uint32_t c = codes[((int64_t)(j >> 2) * n_slots + s) * 4 + (j & 3)];
    w |= c << (8 * u);
  }
  *reinterpret_cast<uint32_t*>(packed + scan_layout::packed_offset(m, n_slots, d * 4, s)) = w;
}

// codes [m][n] -> storage [m/4][n_slots][4] (+ packed) at address[i]; one thread per (i, dword)
__global__ __launch_bounds__(256) void scatter_codes_kernel(const uint8_t* __restrict__ codes,
                                                            const int64_t* __restrict__ address,
                                                            uint8_t* __restrict__ storage,
                                                            uint8_t* __restrict__ packed, int m,
                                                            int64_t n, int64_t n_slots) {
  const int G = m >> 2;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * G) return;
  const int g = (int)(t / n);
  const int64_t i = t - (int64_t)g * n;
  const int64_t s = address[i];
  if (s < 0 || s >= n_slots) return;  // CellContainer.set_data_by_address mask (:238-239)
  uint32_t w = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) w |= (uint32_t)codes[(int64_t)(g * 4 + u) * n + i] << (8 * u);
  reinterpret_cast<uint32_t*>(storage)[(int64_t)g * n_slots + s] = w;
  if (packed) {
    uint32_t pw = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = g * 4 + u;
      const int j = scan_layout::subq_at(m, p, s);
      pw |= (uint32_t)codes[(int64_t)j * n + i] << (8 * u);
    }
    *reinterpret_cast<uint32_t*>(packed + scan_layout::packed_offset(m, n_slots, g * 4, s)) = pw;
  }
}

}  // namespace tpq

using namespace tpq;

extern "C" int tpq_ivfpq_pack_codes(const uint8_t* codes, uint8_t* packed, int64_t n_slots, int m,
                                    int64_t slot_begin, int64_t slot_end, tpq_stream_t stream) {
  TPQ_REQUIRE(codes && packed, "pack_codes: null pointer");
  TPQ_REQUIRE(m >= 4 && m % 4 == 0, "pack_codes: n_subvectors=%d must be a positive multiple of 4", m);
  TPQ_REQUIRE(0 <= slot_begin && slot_begin <= slot_end && slot_end <= n_slots,
              "pack_codes: bad slot range [%lld, %lld) of %lld", (long long)slot_begin,
              (long long)slot_end, (long long)n_slots);
  const int64_t count = slot_end - slot_begin;
  if (count == 0) return TPQ_OK;
  const int64_t total = count * (m / 4);
  hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), codes, packed, n_slots, m, slot_begin,
                     count);
  TPQ_LAUNCH_CHECK("pack_codes_kernel");
  return TPQ_OK;
}

extern "C" int tpq_scatter_codes(const uint8_t* codes, const int64_t* address, uint8_t* storage,
                                 uint8_t* packed, int m, int64_t n, int64_t n_slots,
                                 tpq_stream_t stream) {
  TPQ_REQUIRE(codes && address && storage, "scatter_codes: null pointer");
  TPQ_REQUIRE(m >= 4 && m % 4 == 0, "scatter_codes: n_subvectors=%d must be a positive multiple of 4", m);
  if (n == 0) return TPQ_OK;
  const int64_t total = n * (m / 4);
  hipLaunchKernelGGL(scatter_codes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), codes, address, storage, packed, m, n,
                     n_slots);
  TPQ_LAUNCH_CHECK("scatter_codes_kernel");
  return TPQ_OK;
}
