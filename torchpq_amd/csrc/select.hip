// Row-wise top-k select, smart probing and address->id gather.
//
// tpq_topk_select replaces Top1Select / Top32Select / TopkSelect
// (torchpq/kernels/cuda/top1_select.cu:542, top32_select.cu:484-636, topk_select.cu:662-805,
// dispatch torchpq/fn/Topk.py:43-67): one 64-lane wave per row, register top-k (wave_topk.h).
#include "common.h"
#include "wave_topk.h"

namespace tpq {

constexpr int kSelWaves = 4;

// a2 / b2 non-null: the coarse-probe epilogue of metric.negative_squared_l2_distance
// (torchpq/metric.py:89-96) is applied on the fly -- v = (2*x - a2[row]) - b2[col], the reference's
// order of roundings -- so the three element-wise passes over the [nq, n_cells] GEMM output vanish.
// Optional coarse-probe epilogue (tpq_ivfpq_coarse_probe): the selected columns are cells, so the
// same wave also gathers their list extents (IVFPQIndex.search_cells, index/IVFPQIndex.py:425-426)
// and derives the per-query probe count (smart probing :499-512, or all of them).
struct ProbeEpilogue {
  const int64_t* cell_start_tbl;  // [cols]; nullptr = no epilogue
  const int64_t* cell_size_tbl;
  int64_t* out_cell_start;        // [rows][k]
  int64_t* out_cell_size;
  int64_t* n_probe_list;          // [rows]
  float inv_t;                    // 1 / temperature; <= 0: n_probe_list = k
};

template <int R>
__global__ __launch_bounds__(kSelWaves * 64) void topk_select_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ a2,
                                                                    const float* __restrict__ b2,
                                                                    float* __restrict__ vals,
                                                                    int64_t* __restrict__ idx,
                                                                    int rows, int cols, int k,
                                                                    ProbeEpilogue pe) {
  __shared__ float qv[kSelWaves * 64];
  __shared__ int qi[kSelWaves * 64];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows) return;
  WaveSelector<R> sel;
  sel.init(qv + wave * 64, qi + wave * 64, k);
  const float* __restrict__ xr = x + (int64_t)row * cols;
  const float ra2 = a2 ? a2[row] : 0.f;
  for (int base = 0; base < cols; base += 64) {
    const int c = base + lane;
    const bool valid = c < cols;
    float v = -INFINITY;
    if (valid) {
      v = xr[c];
      if (a2) {
        v = 2.f * v;
        v = v - ra2;
        v = v - b2[c];
      }
      v = v + 0.0f;  // -0.0 -> +0.0 (key order)
    }
    sel.push(valid && (v >= sel.tau), v, c);
  }
  sel.flush();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * 64 + lane;
    if (e < k) {
      const int ci = key_index(sel.top.k[r]);
      const bool pad = ci == kPadIdx;
      vals[(int64_t)row * k + e] = pad ? -INFINITY : key_value(sel.top.k[r]);
      idx[(int64_t)row * k + e] = pad ? -1 : (int64_t)ci;
      if (pe.cell_start_tbl) {
        pe.out_cell_start[(int64_t)row * k + e] = pad ? 0 : pe.cell_start_tbl[ci];
        pe.out_cell_size[(int64_t)row * k + e] = pad ? 0 : pe.cell_size_tbl[ci];
      }
    }
  }
  if (!pe.cell_start_tbl) return;
  if (!(pe.inv_t > 0.f) || k < 2) {
    if (lane == 0) pe.n_probe_list[row] = k;
    return;
  }
  // smart probing on the register-resident sims: element e = r*64 + lane, the assignment (and so
  // the summation order) of smart_probing_kernel below
  float zmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r * 64 + lane < k) zmax = fmaxf(zmax, -sqrtf(fabsf(key_value(sel.top.k[r]))) * pe.inv_t);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, d, 64));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r * 64 + lane < k) sum += expf(-sqrtf(fabsf(key_value(sel.top.k[r]))) * pe.inv_t - zmax);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
  const float inv_log = 1.0f / log2f((float)k);
  float h = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r * 64 + lane < k) {
      const float p = expf(-sqrtf(fabsf(key_value(sel.top.k[r]))) * pe.inv_t - zmax) / sum;
      if (p > 0.f) h -= p * log2f(p) * inv_log;
    }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d, 64);
  if (lane == 0) {
    long long n = (long long)ceilf(h * (float)k);
    n = n < 1 ? 1 : (n > k ? k : n);
    pe.n_probe_list[row] = n;
  }
}

// IVFPQIndex.py:499-512.  One wave per row.
__global__ __launch_bounds__(256) void smart_probing_kernel(const float* __restrict__ sims,
                                                           int64_t* __restrict__ out, int rows,
                                                           int n_probe, float inv_t) {
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* __restrict__ s = sims + (int64_t)row * n_probe;
  float zmax = -INFINITY;
  for (int i = lane; i < n_probe; i += 64) zmax = fmaxf(zmax, -sqrtf(fabsf(s[i])) * inv_t);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, d, 64));
  float sum = 0.f;
  for (int i = lane; i < n_probe; i += 64) sum += expf(-sqrtf(fabsf(s[i])) * inv_t - zmax);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
  const float inv_log = 1.0f / log2f((float)n_probe);
  float h = 0.f;
  for (int i = lane; i < n_probe; i += 64) {
    const float p = expf(-sqrtf(fabsf(s[i])) * inv_t - zmax) / sum;
    if (p > 0.f) h -= p * log2f(p) * inv_log;  // 0*log2(0) := 0 (the reference yields NaN)
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d, 64);
  if (lane == 0) {
    long long n = (long long)ceilf(h * (float)n_probe);
    n = n < 1 ? 1 : (n > n_probe ? n_probe : n);  // always scan the best cell
    out[row] = n;
  }
}

__global__ __launch_bounds__(256) void id_by_address_kernel(const int64_t* __restrict__ a2i,
                                                           int64_t cap,
                                                           const int64_t* __restrict__ adr,
                                                           int64_t* __restrict__ ids, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t a = adr[i];
  ids[i] = (a >= 0 && a < cap) ? a2i[a] : -1;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Coarse similarities sims[q][c] = 2 x_q.C_c - |x_q|^2 - |C_c|^2 (metric.negative_squared_l2_distance,
// torchpq/metric.py:31-98: library GEMM + three element-wise passes) as one fp32-MFMA kernel:
// block = 64 queries x 256 centroids, wave = 32 x 128 (four 32x32 accumulator tiles on
// v_mfma_f32_32x32x2_f32), operands straight from global memory (both matrices are L2-resident:
// 20 B/clk/CU of L1 traffic against 4 MFMAs per k-pair), norms accumulated from the operand
// registers on the way (even-k chain + odd-k chain), epilogue in the reference's rounding order.
// x [d][nq], C [d][n_cells] -> sims [nq][n_cells]
__global__ __launch_bounds__(256) void coarse_sims_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ C,
                                                         float* __restrict__ sims, int d, int nq,
                                                         int n_cells) {
  __shared__ float q2s[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int qb = blockIdx.x * 64 + 32 * (wave & 1);
  const int cb = blockIdx.y * 256 + 128 * (wave >> 1);
  const int q = qb + l31;
  const bool qok = q < nq;
  f32x16 acc[4];
  float b2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    b2[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  }
  float a2 = 0.f;
  // out-of-range rows / columns read row 0 / column 0 instead (always in bounds); an MFMA output
  // depends only on its own A row and B column, and those outputs are never stored
  const float* __restrict__ xq = x + (qok ? q : 0);
  const float* __restrict__ cc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = cb + 32 * t + l31;
    cc[t] = C + (c < n_cells ? c : 0);
  }
  constexpr int KU = 4;  // k-pairs per batch: 20 loads in flight per lane, one batch ahead
  float av[2][KU], bv[2][KU][4];
  auto load_batch = [&](int k0, float (&a)[KU], float (&b)[KU][4]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int64_t k = k0 + 2 * u + half;
      a[u] = xq[k * nq];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[u][t] = cc[t][k * n_cells];
    }
  };
  auto mma_batch = [&](const float (&a)[KU], const float (&b)[KU][4]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      a2 = fmaf(a[u], a[u], a2);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        b2[t] = fmaf(b[u][t], b[u][t], b2[t]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
      }
    }
  };
  const int n_batches = d / (2 * KU);
  if (n_batches > 0) load_batch(0, av[0], bv[0]);
  for (int bt = 0; bt < n_batches; bt += 2) {
    if (bt + 1 < n_batches) load_batch((bt + 1) * 2 * KU, av[1], bv[1]);
    mma_batch(av[0], bv[0]);
    if (bt + 1 >= n_batches) break;
    if (bt + 2 < n_batches) load_batch((bt + 2) * 2 * KU, av[0], bv[0]);
    mma_batch(av[1], bv[1]);
  }
  for (int k0 = n_batches * 2 * KU; k0 < d; k0 += 2) {  // tail of d % 8
    const int k = k0 + half;
    const bool kok = k < d;
    const float a = kok ? xq[(int64_t)k * nq] : 0.f;
    a2 = fmaf(a, a, a2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = kok ? cc[t][(int64_t)k * n_cells] : 0.f;
      b2[t] = fmaf(b, b, b2[t]);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  }
  a2 += __shfl_xor(a2, 32, 64);
#pragma unroll
  for (int t = 0; t < 4; ++t) b2[t] += __shfl_xor(b2[t], 32, 64);
  if (wave < 2 && half == 0) q2s[32 * wave + l31] = a2;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = cb + 32 * t + l31;
    if (c >= n_cells) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int qq = qb + row;
      if (qq < nq) {
        float v = 2.f * acc[t][r];
        v = v - q2s[32 * (wave & 1) + row];
        v = v - b2[t];
        sims[(int64_t)qq * n_cells + c] = v;
      }
    }
  }
}

template <int R>
static int launch_select(const float* x, const float* a2, const float* b2, float* v, int64_t* i,
                         int rows, int cols, int k, hipStream_t st, const ProbeEpilogue& pe) {
  hipLaunchKernelGGL(topk_select_kernel<R>, dim3((rows + kSelWaves - 1) / kSelWaves),
                     dim3(kSelWaves * 64), 0, st, x, a2, b2, v, i, rows, cols, k, pe);
  TPQ_LAUNCH_CHECK("topk_select_kernel");
  return TPQ_OK;
}

}  // namespace tpq

using namespace tpq;

static int select_impl(const float* x, const float* a2, const float* b2, float* vals, int64_t* idx,
                       int rows, int cols, int k, tpq_stream_t stream,
                       const ProbeEpilogue& pe = ProbeEpilogue{});

extern "C" int tpq_topk_select(const float* x, float* vals, int64_t* idx, int rows, int cols, int k,
                               tpq_stream_t stream) {
  return select_impl(x, nullptr, nullptr, vals, idx, rows, cols, k, stream);
}

extern "C" int tpq_coarse_select(const float* dots, const float* a2, const float* b2, float* vals,
                                 int64_t* idx, int rows, int cols, int k, tpq_stream_t stream) {
  TPQ_REQUIRE(a2 && b2, "coarse_select: null norm pointer");
  return select_impl(dots, a2, b2, vals, idx, rows, cols, k, stream);
}

static int select_impl(const float* x, const float* a2, const float* b2, float* vals, int64_t* idx,
                       int rows, int cols, int k, tpq_stream_t stream, const ProbeEpilogue& pe) {
  TPQ_REQUIRE(x && vals && idx, "topk_select: null pointer");
  TPQ_REQUIRE(rows >= 0 && cols >= 1, "topk_select: bad shape [%d, %d]", rows, cols);
  TPQ_REQUIRE(k >= 1 && k <= 1024 && k <= cols, "topk_select: k=%d out of range (cols=%d, max 1024)", k, cols);
  if (rows == 0) return TPQ_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int r = (k + 63) / 64;
  if (r <= 1) return launch_select<1>(x, a2, b2, vals, idx, rows, cols, k, st, pe);
  if (r <= 2) return launch_select<2>(x, a2, b2, vals, idx, rows, cols, k, st, pe);
  if (r <= 4) return launch_select<4>(x, a2, b2, vals, idx, rows, cols, k, st, pe);
  if (r <= 8) return launch_select<8>(x, a2, b2, vals, idx, rows, cols, k, st, pe);
  return launch_select<16>(x, a2, b2, vals, idx, rows, cols, k, st, pe);
}

extern "C" size_t tpq_ivfpq_coarse_probe_workspace_bytes(int nq, int n_cells) {
  if (nq <= 0 || n_cells <= 0) return 0;
  return (size_t)nq * (size_t)n_cells * sizeof(float);
}

extern "C" int tpq_ivfpq_coarse_probe(const float* query, const float* centroids,
                                      const int64_t* cell_start_tbl, const int64_t* cell_size_tbl,
                                      float* topk_sims, int64_t* cells, int64_t* cell_start,
                                      int64_t* cell_size, int64_t* n_probe_list, int d, int nq,
                                      int n_cells, int n_probe, float smart_temperature,
                                      void* workspace, size_t workspace_bytes,
                                      tpq_stream_t stream) {
  TPQ_REQUIRE(query && centroids && cell_start_tbl && cell_size_tbl && topk_sims && cells &&
                  cell_start && cell_size && n_probe_list,
              "ivfpq_coarse_probe: null pointer argument");
  TPQ_REQUIRE(d >= 1 && nq >= 0 && n_cells >= 1, "ivfpq_coarse_probe: bad shape d=%d nq=%d n_cells=%d",
              d, nq, n_cells);
  TPQ_REQUIRE(n_probe >= 1 && n_probe <= n_cells && n_probe <= 1024,
              "ivfpq_coarse_probe: n_probe=%d out of range (n_cells=%d, max 1024)", n_probe, n_cells);
  if (nq == 0) return TPQ_OK;
  const size_t need = tpq_ivfpq_coarse_probe_workspace_bytes(nq, n_cells);
  if (!workspace || workspace_bytes < need) {
    set_error("ivfpq_coarse_probe: workspace too small (%zu < %zu)", workspace_bytes, need);
    return TPQ_ERR_WORKSPACE;
  }
  float* sims = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(coarse_sims_kernel, dim3((nq + 63) / 64, (n_cells + 255) / 256), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), query, centroids, sims, d, nq, n_cells);
  TPQ_LAUNCH_CHECK("coarse_sims_kernel");
  ProbeEpilogue pe{cell_start_tbl, cell_size_tbl, cell_start, cell_size, n_probe_list,
                   smart_temperature > 0.f ? 1.0f / smart_temperature : 0.f};
  return select_impl(sims, nullptr, nullptr, topk_sims, cells, nq, n_cells, n_probe, stream, pe);
}

extern "C" int tpq_smart_probing(const float* topk_sims, int64_t* n_probe_list, int rows,
                                 int n_probe, float temperature, tpq_stream_t stream) {
  TPQ_REQUIRE(topk_sims && n_probe_list, "smart_probing: null pointer");
  TPQ_REQUIRE(n_probe >= 2, "smart_probing: n_probe=%d must be >= 2", n_probe);
  TPQ_REQUIRE(temperature > 0.f, "smart_probing: temperature must be > 0");
  if (rows <= 0) return TPQ_OK;
  hipLaunchKernelGGL(smart_probing_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), topk_sims, n_probe_list, rows, n_probe,
                     1.0f / temperature);
  TPQ_LAUNCH_CHECK("smart_probing_kernel");
  return TPQ_OK;
}

extern "C" int tpq_get_id_by_address(const int64_t* address2id, int64_t capacity,
                                     const int64_t* address, int64_t* ids, int64_t n,
                                     tpq_stream_t stream) {
  TPQ_REQUIRE(address2id && address && ids, "get_id_by_address: null pointer");
  if (n <= 0) return TPQ_OK;
  hipLaunchKernelGGL(id_by_address_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), address2id, capacity, address, ids, n);
  TPQ_LAUNCH_CHECK("id_by_address_kernel");
  return TPQ_OK;
}
