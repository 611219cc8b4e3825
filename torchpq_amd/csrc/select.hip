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
template <int R>
__global__ __launch_bounds__(kSelWaves * 64) void topk_select_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ a2,
                                                                    const float* __restrict__ b2,
                                                                    float* __restrict__ vals,
                                                                    int64_t* __restrict__ idx,
                                                                    int rows, int cols, int k) {
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
    }
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

template <int R>
static int launch_select(const float* x, const float* a2, const float* b2, float* v, int64_t* i,
                         int rows, int cols, int k, hipStream_t st) {
  hipLaunchKernelGGL(topk_select_kernel<R>, dim3((rows + kSelWaves - 1) / kSelWaves),
                     dim3(kSelWaves * 64), 0, st, x, a2, b2, v, i, rows, cols, k);
  TPQ_LAUNCH_CHECK("topk_select_kernel");
  return TPQ_OK;
}

}  // namespace tpq

using namespace tpq;

static int select_impl(const float* x, const float* a2, const float* b2, float* vals, int64_t* idx,
                       int rows, int cols, int k, tpq_stream_t stream);

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
                       int rows, int cols, int k, tpq_stream_t stream) {
  TPQ_REQUIRE(x && vals && idx, "topk_select: null pointer");
  TPQ_REQUIRE(rows >= 0 && cols >= 1, "topk_select: bad shape [%d, %d]", rows, cols);
  TPQ_REQUIRE(k >= 1 && k <= 1024 && k <= cols, "topk_select: k=%d out of range (cols=%d, max 1024)", k, cols);
  if (rows == 0) return TPQ_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int r = (k + 63) / 64;
  if (r <= 1) return launch_select<1>(x, a2, b2, vals, idx, rows, cols, k, st);
  if (r <= 2) return launch_select<2>(x, a2, b2, vals, idx, rows, cols, k, st);
  if (r <= 4) return launch_select<4>(x, a2, b2, vals, idx, rows, cols, k, st);
  if (r <= 8) return launch_select<8>(x, a2, b2, vals, idx, rows, cols, k, st);
  return launch_select<16>(x, a2, b2, vals, idx, rows, cols, k, st);
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
