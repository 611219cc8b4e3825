// Inverted-list bookkeeping kernels (CellContainer.add / remove helpers) and PQ decode.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace tpq {

// ---- get_ioa -------------------------------------------------------------------------------
// ioa[i] = #{ j < i : labels[j] == labels[i] }.  The reference (get_ioa.cu:9-47) gives every
// unique label one thread that walks ALL n labels -- O(n * n_unique).  Here: stable radix sort
// of (label, position) [rocPRIM], then rank inside each run of equal labels -- O(n).
__global__ __launch_bounds__(256) void ioa_prepare_kernel(const int64_t* __restrict__ labels,
                                                         int* __restrict__ keys,
                                                         int* __restrict__ pos, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = (int)labels[i];
  pos[i] = (int)i;
}

// rank of sorted position r inside its run of equal keys: r - (first position holding keys[r]), the first
// position found by binary search in the sorted keys (one launch; the earlier heads + inclusive max-scan +
// rank passes read and wrote the arrays three times, and rocPRIM's scan consults an environment variable)
__global__ __launch_bounds__(256) void ioa_rank_kernel(const int* __restrict__ keys,
                                                      const int* __restrict__ pos,
                                                      int64_t* __restrict__ ioa, int64_t n) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const int key = keys[r];
  int64_t lo = 0, hi = r;  // first index in [0, r] with keys[index] == key (keys ascending)
  if (r > 0 && keys[r - 1] == key) {
    // gallop back first: runs are short against n, so the search stays inside a few cache lines
    int64_t step = 1;
    hi = r - 1;
    while (hi - step >= 0 && keys[hi - step] == key) {
      hi -= step;
      step <<= 1;
    }
    lo = hi - step >= 0 ? hi - step + 1 : 0;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] == key) hi = mid; else lo = mid + 1;
    }
  } else {
    lo = r;
  }
  ioa[pos[r]] = r - lo;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t ioa_temp_bytes(int64_t n) {
  size_t sort_b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_b, (int*)nullptr, (int*)nullptr, (int*)nullptr,
                            (int*)nullptr, (size_t)n, 0, 32, (hipStream_t)0, false);
  return sort_b;
}

// ---- get_write_address ---------------------------------------------------------------------
// the ioa-th empty slot inside [start, start+capacity): one wave per label, 64 slots per step
__global__ __launch_bounds__(256) void write_address_kernel(const uint8_t* __restrict__ is_empty,
                                                           const int64_t* __restrict__ cell_start,
                                                           const int64_t* __restrict__ cell_cap,
                                                           const int64_t* __restrict__ labels,
                                                           const int64_t* __restrict__ ioa,
                                                           int64_t* __restrict__ out, int64_t n_slots,
                                                           int64_t n_labels) {
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= n_labels) return;
  const int64_t lab = labels[w];
  int64_t want = ioa[w];
  const int64_t st = cell_start[lab];
  int64_t end = st + cell_cap[lab];
  if (end > n_slots) end = n_slots;
  int64_t found = -1;
  for (int64_t base = st; base < end; base += 64) {
    const int64_t a = base + lane;
    const bool e = (a < end) && (is_empty[a] == 1);
    const unsigned long long mask = __ballot(e);
    const int cnt = __popcll(mask);
    if (want < cnt) {
      // position of the (want+1)-th set bit
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
      const unsigned long long hit = __ballot(e && rank == (int)want);
      found = base + (__ffsll((long long)hit) - 1);
      break;
    }
    want -= cnt;
  }
  if (lane == 0) out[w] = found;
}

// ---- get_cell_by_address -------------------------------------------------------------------
__global__ __launch_bounds__(256) void cell_by_address_kernel(const int64_t* __restrict__ adr,
                                                             const int64_t* __restrict__ cell_start,
                                                             const int64_t* __restrict__ cell_cap,
                                                             int64_t* __restrict__ out, int64_t n,
                                                             int64_t n_cells) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t a = adr[i];
  int64_t lo = 0, hi = n_cells;  // last cell with start <= a
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (cell_start[mid] <= a) lo = mid + 1; else hi = mid;
  }
  // among cells sharing a start only the last can have capacity > 0, and upper_bound-1 is it
  const int64_t c = lo - 1;
  out[i] = (c >= 0 && a < cell_start[c] + cell_cap[c]) ? c : -1;
}

// ---- pq_decode -----------------------------------------------------------------------------
// out[(j*ds+e)*n + i] = codebook[(j*ds+e)*256 + codes[j*n+i]]; grid (ceil(n/256), m)
__global__ __launch_bounds__(256) void pq_decode_kernel(const float* __restrict__ codebook,
                                                       const uint8_t* __restrict__ codes,
                                                       float* __restrict__ out, int ds, int64_t n) {
  const int j = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = codes[(int64_t)j * n + i];
  for (int e = 0; e < ds; ++e)
    out[((int64_t)j * ds + e) * n + i] = codebook[((int64_t)j * ds + e) * 256 + c];
}

// ---- get_address_by_id, linear form (use_inverse_id_mapping=False) ---------------------------
// replaces get_address_by_id (torchpq/kernels/cuda/get_address_by_id.cu:8-44): every id is compared
// with every stored id -- O(n_ids x capacity), as in the reference; the result is the SMALLEST
// address holding the id (the reference's CPU form, BaseContainer.py:67-77), -1 when absent.
// grid (address chunks, id tiles of 256): a block stages 256 ids in LDS and streams its addresses.
__global__ __launch_bounds__(256) void address_by_id_init_kernel(int64_t* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = INT64_MAX;
}
__global__ __launch_bounds__(256) void address_by_id_kernel(const int64_t* __restrict__ a2i,
                                                           int64_t cap,
                                                           const int64_t* __restrict__ ids,
                                                           int64_t* __restrict__ out, int64_t n_ids,
                                                           int64_t per_block) {
  __shared__ int64_t want[256];
  const int64_t i0 = (int64_t)blockIdx.y * 256;
  const int64_t wi = i0 + threadIdx.x;
  want[threadIdx.x] = wi < n_ids ? ids[wi] : -1;
  __syncthreads();
  const int n_want = (int)((n_ids - i0) < 256 ? (n_ids - i0) : 256);
  const int64_t a0 = (int64_t)blockIdx.x * per_block;
  const int64_t a1 = (a0 + per_block) < cap ? (a0 + per_block) : cap;
  for (int64_t a = a0 + threadIdx.x; a < a1; a += 256) {
    const int64_t v = a2i[a];
    if (v < 0) continue;  // free slot
    for (int j = 0; j < n_want; ++j)
      if (want[j] == v) atomicMin(reinterpret_cast<unsigned long long*>(&out[i0 + j]), (unsigned long long)a);
  }
}
__global__ __launch_bounds__(256) void address_by_id_finish_kernel(int64_t* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && out[i] == INT64_MAX) out[i] = -1;
}

// ---- expand(): re-lay the inverted lists out for larger per-cell capacities ---------------------
// replaces the per-cell torch.cat loop of CellContainer.expand (torchpq/container/CellContainer.py:
// 249-311: O(capacity) work PER expanding cell).  One pass: cell c's slots move from
// [old_start, old_start + old_cap) to [new_start, ...), its new tail is initialised free.
// grid (n_cells, Y): block (c, y) walks slots y*256 + t, step Y*256, of the cell's NEW range.
__global__ __launch_bounds__(256) void grow_cells_kernel(
    const uint32_t* __restrict__ codes, const int64_t* __restrict__ a2i,
    const uint8_t* __restrict__ is_empty, const int64_t* __restrict__ old_start,
    const int64_t* __restrict__ old_cap, const int64_t* __restrict__ new_start,
    const int64_t* __restrict__ new_cap, uint32_t* __restrict__ new_codes,
    int64_t* __restrict__ new_a2i, uint8_t* __restrict__ new_is_empty, int64_t old_slots,
    int64_t new_slots, int G) {
  const int c = blockIdx.x;
  const int64_t os = old_start[c], oc = old_cap[c], ns = new_start[c], nc = new_cap[c];
  for (int64_t s = (int64_t)blockIdx.y * 256 + threadIdx.x; s < nc; s += (int64_t)gridDim.y * 256) {
    const bool had = s < oc;
    new_a2i[ns + s] = had ? a2i[os + s] : -1;
    new_is_empty[ns + s] = had ? is_empty[os + s] : (uint8_t)1;
    for (int g = 0; g < G; ++g)
      new_codes[(int64_t)g * new_slots + ns + s] = had ? codes[(int64_t)g * old_slots + os + s] : 0u;
  }
}

}  // namespace tpq

using namespace tpq;

extern "C" size_t tpq_get_ioa_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return 4 * align256((size_t)n * sizeof(int)) + align256(ioa_temp_bytes(n));
}

extern "C" int tpq_get_ioa(const int64_t* labels, int64_t* ioa, int64_t n, int64_t n_cells,
                           void* workspace, size_t workspace_bytes, tpq_stream_t stream) {
  TPQ_REQUIRE(labels && ioa, "get_ioa: null pointer");
  TPQ_REQUIRE(n >= 0 && n < 0x7fffffffLL, "get_ioa: n=%lld out of range", (long long)n);
  TPQ_REQUIRE(n_cells >= 1 && n_cells < 0x7fffffffLL, "get_ioa: n_cells out of range");
  if (n == 0) return TPQ_OK;
  const size_t need = tpq_get_ioa_workspace_bytes(n);
  if (!workspace || workspace_bytes < need) {
    set_error("get_ioa: workspace too small (%zu < %zu)", workspace_bytes, need);
    return TPQ_ERR_WORKSPACE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* p = reinterpret_cast<char*>(workspace);
  const size_t arr = align256((size_t)n * sizeof(int));
  int* keys_in = reinterpret_cast<int*>(p);
  int* keys_out = reinterpret_cast<int*>(p + arr);
  int* pos_in = reinterpret_cast<int*>(p + 2 * arr);
  int* pos_out = reinterpret_cast<int*>(p + 3 * arr);
  void* temp = p + 4 * arr;
  size_t temp_bytes = ioa_temp_bytes(n);
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(ioa_prepare_kernel, dim3(grid), dim3(256), 0, st, labels, keys_in, pos_in, n);
  TPQ_LAUNCH_CHECK("ioa_prepare_kernel");
  int bits = 1;
  while (bits < 32 && (1LL << bits) < n_cells) ++bits;
  int rc = check_hip(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, pos_in, pos_out,
                                               (size_t)n, 0, bits, st, false),
                     "get_ioa radix_sort_pairs");
  if (rc) return rc;
  hipLaunchKernelGGL(ioa_rank_kernel, dim3(grid), dim3(256), 0, st, keys_out, pos_out, ioa, n);
  TPQ_LAUNCH_CHECK("ioa_rank_kernel");
  return TPQ_OK;
}

extern "C" int tpq_get_write_address(const uint8_t* is_empty, const int64_t* cell_start,
                                     const int64_t* cell_capacity, const int64_t* labels,
                                     const int64_t* ioa, int64_t* write_address, int64_t n_slots,
                                     int64_t n_labels, tpq_stream_t stream) {
  TPQ_REQUIRE(is_empty && cell_start && cell_capacity && labels && ioa && write_address,
              "get_write_address: null pointer");
  if (n_labels <= 0) return TPQ_OK;
  hipLaunchKernelGGL(write_address_kernel, dim3((unsigned)((n_labels + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), is_empty, cell_start, cell_capacity,
                     labels, ioa, write_address, n_slots, n_labels);
  TPQ_LAUNCH_CHECK("write_address_kernel");
  return TPQ_OK;
}

extern "C" int tpq_get_cell_by_address(const int64_t* address, const int64_t* cell_start,
                                       const int64_t* cell_capacity, int64_t* cells,
                                       int64_t n_address, int64_t n_cells, tpq_stream_t stream) {
  TPQ_REQUIRE(address && cell_start && cell_capacity && cells, "get_cell_by_address: null pointer");
  if (n_address <= 0) return TPQ_OK;
  hipLaunchKernelGGL(cell_by_address_kernel, dim3((unsigned)((n_address + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), address, cell_start, cell_capacity, cells,
                     n_address, n_cells);
  TPQ_LAUNCH_CHECK("cell_by_address_kernel");
  return TPQ_OK;
}

extern "C" int tpq_pq_decode(const float* codebook, const uint8_t* codes, float* out, int m, int ds,
                             int64_t n, tpq_stream_t stream) {
  TPQ_REQUIRE(codebook && codes && out, "pq_decode: null pointer");
  TPQ_REQUIRE(m >= 1 && m <= 65535 && ds >= 1, "pq_decode: bad shape m=%d ds=%d", m, ds);
  if (n <= 0) return TPQ_OK;
  hipLaunchKernelGGL(pq_decode_kernel, dim3((unsigned)((n + 255) / 256), m), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), codebook, codes, out, ds, n);
  TPQ_LAUNCH_CHECK("pq_decode_kernel");
  return TPQ_OK;
}

extern "C" int tpq_get_address_by_id(const int64_t* address2id, int64_t capacity, const int64_t* ids,
                                     int64_t* address, int64_t n_ids, tpq_stream_t stream) {
  TPQ_REQUIRE(capacity >= 0 && n_ids >= 0, "get_address_by_id: bad sizes");
  if (n_ids == 0) return TPQ_OK;
  TPQ_REQUIRE(ids && address && (address2id || capacity == 0), "get_address_by_id: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = (unsigned)((n_ids + 255) / 256);
  hipLaunchKernelGGL(address_by_id_init_kernel, dim3(nb), dim3(256), 0, st, address, n_ids);
  TPQ_LAUNCH_CHECK("address_by_id_init_kernel");
  if (capacity > 0) {
    TPQ_REQUIRE(nb <= 65535, "get_address_by_id: too many ids for one call (%lld)", (long long)n_ids);
    int64_t chunks = (capacity + 4095) / 4096;
    if (chunks > 4096) chunks = 4096;
    const int64_t per_block = ((capacity + chunks - 1) / chunks + 255) / 256 * 256;
    hipLaunchKernelGGL(address_by_id_kernel, dim3((unsigned)((capacity + per_block - 1) / per_block), nb),
                       dim3(256), 0, st, address2id, capacity, ids, address, n_ids, per_block);
    TPQ_LAUNCH_CHECK("address_by_id_kernel");
  }
  hipLaunchKernelGGL(address_by_id_finish_kernel, dim3(nb), dim3(256), 0, st, address, n_ids);
  TPQ_LAUNCH_CHECK("address_by_id_finish_kernel");
  return TPQ_OK;
}

extern "C" int tpq_grow_cells(const uint8_t* storage, const int64_t* address2id, const uint8_t* is_empty,
                              const int64_t* old_start, const int64_t* old_capacity,
                              const int64_t* new_start, const int64_t* new_capacity,
                              uint8_t* new_storage, int64_t* new_address2id, uint8_t* new_is_empty,
                              int64_t old_slots, int64_t new_slots, int n_cells, int m,
                              tpq_stream_t stream) {
  TPQ_REQUIRE(storage && address2id && is_empty && old_start && old_capacity && new_start &&
                  new_capacity && new_storage && new_address2id && new_is_empty,
              "grow_cells: null pointer");
  TPQ_REQUIRE(m >= 4 && m % 4 == 0, "grow_cells: n_subvectors=%d must be a positive multiple of 4", m);
  TPQ_REQUIRE(n_cells >= 1 && old_slots >= 0 && new_slots >= old_slots, "grow_cells: bad sizes");
  if (new_slots == 0) return TPQ_OK;
  int64_t y = (new_slots / n_cells + 2047) / 2048;
  if (y < 1) y = 1;
  if (y > 256) y = 256;
  hipLaunchKernelGGL(grow_cells_kernel, dim3((unsigned)n_cells, (unsigned)y), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint32_t*>(storage),
                     address2id, is_empty, old_start, old_capacity, new_start, new_capacity,
                     reinterpret_cast<uint32_t*>(new_storage), new_address2id, new_is_empty, old_slots,
                     new_slots, m / 4);
  TPQ_LAUNCH_CHECK("grow_cells_kernel");
  return TPQ_OK;
}
