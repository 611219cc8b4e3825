// Row-wise top-k select, smart probing and address->id gather.
//
// tpq_topk_select replaces Top1Select / Top32Select / TopkSelect
// (torchpq/kernels/cuda/top1_select.cu:542, top32_select.cu:484-636, topk_select.cu:662-805,
// dispatch torchpq/fn/Topk.py:43-67): one 64-lane wave per row, register top-k (wave_topk.h).
#include "common.h"
#include "probe_fast.h"
#include "wave_topk.h"

namespace tpq {

constexpr int kSelWaves = 4;

// a2 / b2 non-null: the coarse-probe epilogue of metric.negative_squared_l2_distance
// (torchpq/metric.py:89-96) is applied on the fly -- v = (2*x - a2[row]) - b2[col], the reference's
// order of roundings -- so the three element-wise passes over the [nq, n_cells] GEMM output vanish.
// Optional coarse-probe epilogue (tpq_ivfpq_coarse_probe): the selected columns are cells, so the
// same wave also gathers their list extents (IVFPQIndex.search_cells, index/IVFPQIndex.py:425-426)
// and derives the per-query probe count (smart probing :499-512, or all of them).
// Optional two-level select: gmax[row][g] = max of the row over columns [128 g, 128 g + 128) (written
// by coarse_sims_kernel).  The k-th largest group maximum is a lower bound of the k-th largest
// element (the k largest group maxima are k distinct elements), so only groups whose maximum
// reaches it can hold a member of the top-k: with n_probe = 8 of 16 384 cells the row select reads
// ~8 % of the row.  The result is the same total order (value desc, column asc) as the full scan.
struct GroupFilter {
  const float* gmax;  // [rows][n_groups]; nullptr = scan every column
  int n_groups;
};

struct ProbeEpilogue {
  const int64_t* cell_start_tbl;  // [cols]; nullptr = no epilogue
  const int64_t* cell_size_tbl;
  int64_t* out_cell_start;        // [rows][k]
  int64_t* out_cell_size;
  int64_t* n_probe_list;          // [rows]
  float inv_t;                    // 1 / temperature; <= 0: n_probe_list = k
};

template <int R>
__device__ __forceinline__ void write_row(const WaveTopK<R>& top, float* __restrict__ vals, int64_t* __restrict__ idx,
                                          int row, int k, const ProbeEpilogue& pe);

// one wave selects row `row` (its values at xr[0 .. cols), global memory or LDS) -- the body of
// topk_select_kernel and of probe_small_kernel
template <int R>
__device__ __forceinline__ void select_row(float* qvw, int* qiw, const float* xr, const float* __restrict__ a2,
                                           const float* __restrict__ b2, float* __restrict__ vals,
                                           int64_t* __restrict__ idx, int row, int cols, int k,
                                           const ProbeEpilogue& pe, const GroupFilter& gf) {
  const int lane = lane_id();
  WaveSelector<R> sel;
  sel.init(qvw, qiw, k);
  const float ra2 = a2 ? a2[row] : 0.f;
  if (gf.gmax) {
    // phase 1: the k-th largest group maximum
    const float* __restrict__ gm = gf.gmax + (int64_t)row * gf.n_groups;
    for (int base = 0; base < gf.n_groups; base += 64) {
      const int g = base + lane;
      const float v = g < gf.n_groups ? gm[g] + 0.0f : -INFINITY;
      sel.push(g < gf.n_groups && (v >= sel.tau), v, g);
    }
    sel.flush();
    const float tau0 = sel.top.kth_value(k);  // -inf while there are fewer than k groups
    sel.init(qvw, qiw, k);
    // phase 2: only the groups that can hold a member of the top-k, four (eight loads) at a time
    for (int base = 0; base < gf.n_groups; base += 64) {
      const int g = base + lane;
      const bool hot = g < gf.n_groups && (gm[g] >= tau0);
      unsigned long long mask = __ballot(hot);
      while (mask != 0ull) {
        int gs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          gs[u] = -1;
          if (mask != 0ull) {
            gs[u] = base + (int)__builtin_ctzll(mask);
            mask &= mask - 1ull;
          }
        }
        float va[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = gs[u >> 1] * 128 + 64 * (u & 1) + lane;
          va[u] = (gs[u >> 1] >= 0 && c < cols) ? xr[c] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (gs[u >> 1] >= 0) {  // wave-uniform
            const int c = gs[u >> 1] * 128 + 64 * (u & 1) + lane;
            const bool valid = c < cols;
            const float v = va[u] + 0.0f;
            sel.push(valid && (v >= sel.tau), v, c);
          }
        }
      }
    }
  } else {
    // kSelAhead 64-column groups are loaded before any of them is pushed: with one load per
    // iteration a wave waits out a full memory latency per 256 bytes (1.9 TB/s on a
    // [10 000 x 16 384] matrix); 16 waves x 4 KiB in flight per CU cover the latency
    constexpr int kSelAhead = 16;
    for (int base = 0; base < cols; base += 64 * kSelAhead) {
      float va[kSelAhead];
  #pragma unroll
      for (int u = 0; u < kSelAhead; ++u) {
        const int c = base + 64 * u + lane;
        va[u] = c < cols ? xr[c] : -INFINITY;
      }
  #pragma unroll
      for (int u = 0; u < kSelAhead; ++u) {
        const int c = base + 64 * u + lane;
        if (base + 64 * u < cols) {  // wave-uniform
          const bool valid = c < cols;
          float v = va[u];
          if (valid) {
            if (a2) {
              v = 2.f * v;
              v = v - ra2;
              v = v - b2[c];
            }
            v = v + 0.0f;  // -0.0 -> +0.0 (key order)
          }
          sel.push(valid && (v >= sel.tau), v, c);
        }
      }
    }
  }
  sel.flush();
  write_row<R>(sel.top, vals, idx, row, k, pe);
}

// the selected row: values, columns and -- coarse probe -- the cells' extents and the probe count
template <int R>
__device__ __forceinline__ void write_row(const WaveTopK<R>& top, float* __restrict__ vals, int64_t* __restrict__ idx,
                                          int row, int k, const ProbeEpilogue& pe) {
  const int lane = lane_id();
  struct {
    const WaveTopK<R>& top;
  } sel{top};
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

template <int R>
__global__ __launch_bounds__(kSelWaves * 64) void topk_select_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ a2,
                                                                    const float* __restrict__ b2,
                                                                    float* __restrict__ vals,
                                                                    int64_t* __restrict__ idx,
                                                                    int rows, int cols, int k,
                                                                    ProbeEpilogue pe, GroupFilter gf) {
  __shared__ float qv[kSelWaves * 64];
  __shared__ int qi[kSelWaves * 64];
  const int wave = threadIdx.x >> 6;
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows) return;
  select_row<R>(qv + wave * 64, qi + wave * 64, x + (int64_t)row * cols, a2, b2, vals, idx, row, cols, k, pe, gf);
}

// The coarse step's row select on FAST similarities (probe_fast.h): one wave per query.
//   1. the k best fast values of the row, kept with a margin: everything within `band` = 2 delta' of the running
//      k-th best is admitted and the list holds 64 R > k entries (the group filter works on fast values too: a group
//      is read when its maximum reaches the k-th largest group maximum minus the band);
//   2. every list entry within the band of the k-th best fast value is a CANDIDATE: the exact top-k is among them
//      (|f' - e'| <= delta' for every cell).  A lane evaluates its candidate with the fp32 kernels' own arithmetic --
//      acc = fma chain over ascending k of C[k][c] x[k], v = ((2 acc) - |x|^2) - |C|^2 -- from the centroid's row copy;
//   3. the candidates are re-ranked by (exact value desc, cell asc) and the best k written: coarse_sims_kernel +
//      topk_select_kernel's output, bit for bit.
// A list full of candidates (an entry may have been evicted), or band = +inf (queries / centroids beyond the fp16 scale):
// the wave evaluates ALL cells of its query exactly -- slow, and normally never taken.
constexpr int kCandCap = 256;  // candidate cells a wave keeps without selecting (direct path)
constexpr int kHotCap = 512;   // hot groups a wave lists

template <int R>
__global__ __launch_bounds__(kSelWaves * 64) void probe_select_fast_kernel(ProbeFastBuffers fb, const float* __restrict__ x,
                                                                          float* __restrict__ vals,
                                                                          int64_t* __restrict__ idx, int d, int nq,
                                                                          int n_cells, int k, ProbeEpilogue pe) {
  __shared__ float qv[kSelWaves * 64];
  __shared__ int qi[kSelWaves * 64];
  __shared__ float xq_all[kSelWaves * 128];
  __shared__ int cand[kSelWaves * kCandCap];
  __shared__ int hot[kSelWaves * kHotCap];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= nq) return;
  float* xq = xq_all + wave * 128;
  // everything the wave needs first, issued together: its query's row (d <= 128 floats), |x|^2, band, scale and the
  // first group maxima
  const float4 xrow = lane * 4 < d ? reinterpret_cast<const float4*>(fb.xt + (int64_t)row * fb.xt_stride)[lane]
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  const float q2 = fb.q2[row];
  const float band0 = fb.band[row];
  const float qs = fb.qscale[row];
  const float* __restrict__ gm = fb.gmax + (int64_t)row * fb.n_groups;   // f' (fp32): scaled on the fly
  float gm0[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) gm0[u] = 64 * u + lane < fb.n_groups ? gm[64 * u + lane] : -INFINITY;
  if (lane * 4 < d) reinterpret_cast<float4*>(xq)[lane] = xrow;  // (d % 4 != 0: the row copy is zero-padded to xt_stride)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  auto exact = [&](int c) -> float {  // the fp32 kernels' value of (query, cell c)
    const float4* __restrict__ cr = reinterpret_cast<const float4*>(fb.ct + (int64_t)c * d);
    float acc = 0.f;
    int t = 0;
    for (; t + 32 <= d; t += 32) {  // (each candidate's row is a lane's own: eight loads in flight, four round trips at d = 128)
      float4 y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) y[u] = cr[(t >> 2) + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = fmaf(y[u].x, xq[t + 4 * u], acc);
        acc = fmaf(y[u].y, xq[t + 4 * u + 1], acc);
        acc = fmaf(y[u].z, xq[t + 4 * u + 2], acc);
        acc = fmaf(y[u].w, xq[t + 4 * u + 3], acc);
      }
    }
    for (; t + 16 <= d; t += 16) {
      float4 y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) y[u] = cr[(t >> 2) + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc = fmaf(y[u].x, xq[t + 4 * u], acc);
        acc = fmaf(y[u].y, xq[t + 4 * u + 1], acc);
        acc = fmaf(y[u].z, xq[t + 4 * u + 2], acc);
        acc = fmaf(y[u].w, xq[t + 4 * u + 3], acc);
      }
    }
    for (; t < d; ++t) acc = fmaf(fb.ct[(int64_t)c * d + t], xq[t], acc);
    float v = 2.f * acc;
    v = v - q2;
    v = v - fb.c2[c];
    return v + 0.0f;
  };
#ifdef TPQ_SELECT_STOP
#define TPQ_STOP_AT(n, val) if (TPQ_SELECT_STOP == n) { if (lane == 0) vals[(int64_t)row * k] = (val); return; }
#else
#define TPQ_STOP_AT(n, val)
#endif
  TPQ_STOP_AT(1, q2)
  WaveTopK<R> ex;
  bool slow = !(band0 < INFINITY);
  if (!slow) {
    WaveSelector<R> sel;
    sel.init(qv + wave * 64, qi + wave * 64, k);
    sel.margin = band0;
    const _Float16* __restrict__ xr = fb.sims + (int64_t)row * n_cells;  // stored units: f' x qs, fp16
    const uint32_t* __restrict__ xr2 = reinterpret_cast<const uint32_t*>(xr);  // (rows are 64-byte aligned: n_cells % 32 == 0)
    // phase 1: the k-th largest group maximum (a lower bound of the k-th largest fast value)
    for (int base = 0; base < fb.n_groups; base += 64) {
      const int g = base + lane;
      const float gv = base < 256 ? gm0[(base >> 6) & 3] : (g < fb.n_groups ? gm[g] : -INFINITY);
      const float v = g < fb.n_groups ? gv * qs + 0.0f : -INFINITY;
      sel.push(g < fb.n_groups && (v >= sel.tau - band0), v, g);
    }
    sel.flush();
    // The stored values are fp16: u = f' x qs rounded to nearest, |stored - u| <= 2^-11 |u| (+ 2^-25 where the result is
    // subnormal).  A cell that belongs to the exact top k has u in [G_k - band0, M_1] (G_k the k-th largest group
    // maximum -- a lower bound of the k-th largest u --, M_1 the largest; both unrounded), so its stored value is within
    // eps = 2^-11 (max(|M_1|, |G_k|) + band0) of u; and the k-th largest stored value is within eps of the k-th largest u
    // (rounding is monotone, the k-th largest u lies in [G_k, M_1]).  Band in stored values: band0 + 2 eps.  |u| < 2^15
    // by the choice of qs: eps <= 16 whatever the row holds (fewer than k groups: G_k = -inf).
    const float gk = sel.top.kth_value(k);
    const float mag = fmaxf(fabsf(sel.top.kth_value(1)), fabsf(gk)) + band0;
    const float eps = fminf(16.f, mag * 4.8828125e-4f) * 1.001f + 5.9604645e-8f;
    const float band = band0 + 2.f * eps;
    const float tau0 = gk - band;  // -inf while there are fewer than k groups
    TPQ_STOP_AT(2, tau0)
    // the hot groups -- those whose maximum reaches tau0 -- as a list in LDS, then their cells two per lane (a dword of
    // the fp16 row; a 32-cell group is 16 lanes of a load, a 128-cell group all 64), eight loads a round: the walk is a
    // chain of memory round trips and there are as many of them as rounds
    int* hl = hot + wave * kHotCap;
    int n_hot = 0;  // wave-uniform
    for (int base = 0; base < fb.n_groups; base += 64) {
      const int g = base + lane;
      const float gv = base < 256 ? gm0[(base >> 6) & 3] : (g < fb.n_groups ? gm[g] : -INFINITY);
      const bool is_hot = g < fb.n_groups && (gv * qs >= tau0);
      const unsigned long long b = __ballot(is_hot);
      const int pos = n_hot + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
      if (is_hot && pos < kHotCap) hl[pos] = g;
      n_hot += __popcll(b);
    }
    const bool hot_listed = n_hot <= kHotCap;  // (more groups than the list holds: only beyond 65 536 cells; exact then)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (the wave's own LDS appends)
    const int lpg_shift = fb.gshift - 1;                    // lanes per group: a lane holds two cells
    const int sub = lane >> lpg_shift, lig = lane & ((1 << lpg_shift) - 1);
    const int gpl = 64 >> lpg_shift;                        // groups per load
    auto walk = [&](auto&& consume, auto&& go_on) {
      for (int h0 = 0; h0 < n_hot && go_on(); h0 += 8 * gpl) {
        uint32_t vw[8];
        int cb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int hi = h0 + u * gpl + sub;
          const int g = hi < n_hot ? hl[hi] : -1;
          const int c = (g << fb.gshift) + 2 * lig;   // (n_cells is even: a pair is inside the row or beyond it)
          const bool ok = g >= 0 && c < n_cells;
          vw[u] = ok ? xr2[c >> 1] : 0xfc00fc00u;      // (-inf, -inf)
          cb[u] = ok ? c : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (h0 + u * gpl < n_hot) {  // wave-uniform
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            const f16x2 hv = __builtin_bit_cast(f16x2, vw[u]);
            consume(cb[u] >= 0, cb[u], (float)hv[0]);
            consume(cb[u] >= 0, cb[u] + 1, (float)hv[1]);
          }
        }
      }
    };
    // phase 2, direct: EVERY cell of a hot group whose stored value reaches tau0 is kept -- a superset of the candidates
    // (cut >= tau0: the k-th largest stored value is not below G_k - eps) that costs a ballot and an LDS append per 64
    // cells instead of the selector's queue, sorts and merges; the exact top k is among them whatever else is, so the
    // exact values of all of them, sorted once, are the answer.  More than kCandCap of them (k close to or beyond the
    // number of groups: G_k is a poor bound or none) and the selector path below finds the cut itself.
    int* cl = cand + wave * kCandCap;
    int n_cand = 0;  // wave-uniform
    bool direct = hot_listed && gk > -INFINITY && 2 * k <= fb.n_groups;  // (k-th of fewer than 2 k maxima: too low a bound to try)
    if (direct) {
      walk(
          [&](bool valid, int c, float v) {
            const bool keep = valid && v >= tau0;
            const unsigned long long b = __ballot(keep);
            const int pos = n_cand + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
            if (keep && pos < kCandCap) cl[pos] = c;
            n_cand += __popcll(b);
          },
          [&]() { return n_cand <= kCandCap; });
      direct = n_cand <= kCandCap;
    }
    TPQ_STOP_AT(3, (float)n_cand)
    if (direct) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (the wave's own LDS appends)
      ex.init();
      for (int base = 0; base < n_cand; base += 64) {
        const bool want = base + lane < n_cand;
        const int c = want ? cl[base + lane] : 0;
        const float e = want ? exact(c) : -INFINITY;
        ex.insert_unsorted(want ? make_key(e, c) : pad_key());
      }
      TPQ_STOP_AT(4, ex.kth_value(1))
    } else if (!hot_listed) {
      slow = true;
    } else {
    sel.init(qv + wave * 64, qi + wave * 64, k);
    sel.margin = band;
    // phase 2 through the selector: the k best stored values with their band
    walk([&](bool valid, int c, float v0) {
           const float v = v0 + 0.0f;
           sel.push(valid && (v >= sel.tau - band), v, c);
         },
         [&]() { return true; });
    sel.flush();
    const float cut = sel.top.kth_value(k) - band;
    const Key last = readlane_key(sel.top.k[R - 1], 63);
    if (key_index(last) != kPadIdx && key_value(last) >= cut) slow = true;  // a full list of candidates: wave-uniform
    if (!slow) {
      ex.init();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int c = key_index(sel.top.k[r]);
        const bool want = c != kPadIdx && key_value(sel.top.k[r]) >= cut;
        if (__ballot(want) == 0ull) break;  // sorted by fast value: nothing further down qualifies
        const float e = want ? exact(c) : -INFINITY;
        ex.insert_unsorted(want ? make_key(e, c) : pad_key());
      }
    }
    }
  }
  if (slow) {  // every cell, exactly
    WaveSelector<R> sel;
    sel.init(qv + wave * 64, qi + wave * 64, k);
    for (int base = 0; base < n_cells; base += 64) {
      const int c = base + lane;
      const float v = c < n_cells ? exact(c) : -INFINITY;
      sel.push(c < n_cells && (v >= sel.tau), v, c);
    }
    sel.flush();
    ex = sel.top;
  }
  write_row<R>(ex, vals, idx, row, k, pe);
}

// Small batches (tpq_ivfpq_coarse_probe, nq <= kProbeSmallMaxQ): the whole coarse step of a query in ONE
// block -- its sims row computed into LDS, selected by wave 0 -- instead of the sims kernel + the select
// kernel (at one query the launch gaps and the second kernel's start-up are most of the 28 us).
// One thread per cell: dot, |C|^2 and (every thread) |x|^2 as ascending-k fmaf chains, v = (2 dot - |x|^2)
// - |C|^2: the arithmetic of coarse_sims_kernel and oracle_coarse_sims, bit for bit.  The chains are
// sequential in k, the loads are not: 16 in flight per thread.
constexpr int kProbeSmallThreads = 1024;
constexpr int kProbeSmallMaxQ = 256;
constexpr int kProbeSmallMaxCells = 8192;   // sims row in LDS (32 KiB)
constexpr int kProbeSmallMaxD = 1024;       // query in LDS

template <int R>
__global__ __launch_bounds__(kProbeSmallThreads) void probe_small_kernel(const float* __restrict__ x,
                                                                        const float* __restrict__ C,
                                                                        float* __restrict__ vals,
                                                                        int64_t* __restrict__ idx, int d, int nq,
                                                                        int n_cells, int k, ProbeEpilogue pe) {
  __shared__ float row_s[kProbeSmallMaxCells];
  __shared__ float xq[kProbeSmallMaxD];
  __shared__ float qv[64];
  __shared__ int qi[64];
  const int q = blockIdx.x;
  for (int t = threadIdx.x; t < d; t += kProbeSmallThreads) xq[t] = x[(int64_t)t * nq + q];
  __syncthreads();
  float q2 = 0.f;
  for (int t = 0; t < d; ++t) q2 = fmaf(xq[t], xq[t], q2);
  for (int c = threadIdx.x; c < n_cells; c += kProbeSmallThreads) {
    const float* __restrict__ p = C + c;
    float acc = 0.f, c2 = 0.f;
    int t = 0;
    // 64 loads in flight per thread (the chains are sequential in k, the loads are not): at one query
    // the block is alone on the chip and the 512 KiB of centroids come from L2 / the Infinity Cache --
    // with 16 in flight the eight round trips were most of the kernel's 30 us
    for (; t + 64 <= d; t += 64) {
      float y[64];
#pragma unroll
      for (int u = 0; u < 64; ++u) y[u] = p[(int64_t)(t + u) * n_cells];
#pragma unroll
      for (int u = 0; u < 64; ++u) {
        acc = fmaf(y[u], xq[t + u], acc);
        c2 = fmaf(y[u], y[u], c2);
      }
    }
    for (; t + 16 <= d; t += 16) {
      float y[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) y[u] = p[(int64_t)(t + u) * n_cells];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc = fmaf(y[u], xq[t + u], acc);
        c2 = fmaf(y[u], y[u], c2);
      }
    }
    for (; t < d; ++t) {
      const float y = p[(int64_t)t * n_cells];
      acc = fmaf(y, xq[t], acc);
      c2 = fmaf(y, y, c2);
    }
    float v = 2.f * acc;
    v = v - q2;
    v = v - c2;
    row_s[c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) select_row<R>(qv, qi, row_s, nullptr, nullptr, vals, idx, q, n_cells, k, pe, GroupFilter{nullptr, 0});
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
// torchpq/metric.py:31-98: library GEMM + three element-wise passes) as one fp32-MFMA kernel, built
// like max_sim_kernel (kmeans.hip): a block owns 128 QUERIES (4 waves x 32 MFMA columns, operand
// in registers, prefetched one k-slab ahead) and walks centroid chunks of 256 MFMA rows whose
// 16-row k-slabs are double-buffered in LDS (global -> registers while the previous slab's 8 x 8
// MFMAs run -> the other buffer, one barrier per slab).  |C|^2 is accumulated from the values each
// thread stages (its centroid, every slab, ascending k), |x|^2 by each lane for its own query.
// With the queries on the lanes
//   * the maximum of a query's sims over a 128-centroid group is an in-lane reduction over
//     accumulator registers -> gmax[q][group], which lets the row select skip every group that
//     cannot hold a member of the top-n_probe (GroupFilter above);
//   * a tile's sims leave through a 32 x 33 LDS transpose per wave, so that a half-wave still
//     stores 128 contiguous bytes of a sims row.
// In the reference's own benchmark grid (IVF4096 / IVF16384, n_probe 1..128) this step is 40-85 %
// of a search, not the scan.
// x [d][nq], C [d][n_cells] -> sims [nq][n_cells], gmax [nq][ceil(n_cells/128)]
// grid (ceil(nq/128), centroid-chunk groups)
constexpr int kCsRows = 256;  // centroids per chunk (8 MFMA row tiles = 2 groups of 128)
constexpr int kCsKC = 16;     // k rows per LDS slab
constexpr int kCsSlab = kCsKC * kCsRows;

__global__ __launch_bounds__(256, 2) void coarse_sims_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ C,
                                                            float* __restrict__ sims, int d, int nq,
                                                            int n_cells, int chunks_per_block,
                                                            float* __restrict__ gmax, int n_groups) {
  __shared__ float cs[2 * kCsSlab];   // [2][kCsKC][kCsRows]
  __shared__ float c2s[kCsRows];
  __shared__ float tr[4 * 32 * 33];   // per wave: 32 queries x (32 + 1) centroids
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int qw = blockIdx.x * 128 + wave * 32;   // first query of this wave
  const int q = qw + l31;                        // this lane's query
  const bool qvalid = q < nq;
  const float* __restrict__ xq = x + (qvalid ? q : 0);
  float* trw = tr + wave * 32 * 33;

  float q2 = 0.f;  // |x_q|^2, one ascending-k chain, 16 loads in flight per step
  {
    const float* __restrict__ p = xq;
    int k = 0;
    for (; k + 16 <= d; k += 16) {
      float y[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) y[u] = p[(int64_t)u * nq];
#pragma unroll
      for (int u = 0; u < 16; ++u) q2 = fmaf(y[u], y[u], q2);
      p += 16 * (int64_t)nq;
    }
    for (; k < d; ++k) {
      q2 = fmaf(*p, *p, q2);
      p += nq;
    }
  }

  const int n_slabs = (d + kCsKC - 1) / kCsKC;
  const int chunk0 = blockIdx.y * chunks_per_block;
  for (int ch = chunk0; ch < chunk0 + chunks_per_block; ++ch) {
    const int c0 = ch * kCsRows;
    if (c0 >= n_cells) break;
    const int nc = (n_cells - c0) < kCsRows ? (n_cells - c0) : kCsRows;
    const bool cv = (int)threadIdx.x < nc;  // this thread's centroid row of the chunk exists
    const float* __restrict__ Cc = C + c0 + (cv ? (int)threadIdx.x : 0);
    float rs[kCsKC], yc[kCsKC / 2], yn[kCsKC / 2];
    float csq = 0.f;
    auto load_slab = [&](int kb) {
      const float* __restrict__ p = Cc + (int64_t)kb * n_cells;
#pragma unroll
      for (int u = 0; u < kCsKC; ++u) {
        rs[u] = (cv && kb + u < d) ? *p : 0.f;
        p += n_cells;
      }
    };
    auto square_slab = [&]() {
#pragma unroll
      for (int u = 0; u < kCsKC; ++u) csq = fmaf(rs[u], rs[u], csq);
    };
    auto store_slab = [&](float* dst) {
#pragma unroll
      for (int u = 0; u < kCsKC; ++u) dst[u * kCsRows + threadIdx.x] = rs[u];
    };
    auto load_y = [&](int kb, float (&y)[kCsKC / 2]) {
      const float* __restrict__ p = xq + (int64_t)(kb + half) * nq;
#pragma unroll
      for (int j = 0; j < kCsKC / 2; ++j) {
        y[j] = (qvalid && kb + 2 * j + half < d) ? *p : 0.f;
        p += 2 * (int64_t)nq;
      }
    };
    load_slab(0);
    load_y(0, yc);
    __syncthreads();  // every wave finished the previous chunk (reads of cs and c2s)
    square_slab();
    // (rows past the last centroid get |C|^2 = +inf: their sims come out as -inf and drop out of
    // the group maxima without a per-element predicate)
    if (n_slabs == 1) c2s[threadIdx.x] = cv ? csq : INFINITY;
    store_slab(cs);
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();
    for (int sb = 0; sb < n_slabs; ++sb) {
      const float* cur = cs + (sb & 1) * kCsSlab;
      const bool more = sb + 1 < n_slabs;
      if (more) {
        load_slab((sb + 1) * kCsKC);
        load_y((sb + 1) * kCsKC, yn);
      }
#pragma unroll
      for (int j = 0; j < kCsKC / 2; ++j) {
        const float* crow = cur + (2 * j + half) * kCsRows + l31;  // A operand [row=centroid][k]
#pragma unroll
        for (int t = 0; t < 8; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(crow[t * 32], yc[j], acc[t], 0, 0, 0);
      }
      if (more) {
        square_slab();
        if (sb + 2 == n_slabs) c2s[threadIdx.x] = cv ? csq : INFINITY;
        store_slab(cs + ((sb + 1) & 1) * kCsSlab);
#pragma unroll
        for (int j = 0; j < kCsKC / 2; ++j) yc[j] = yn[j];
      }
      __syncthreads();
    }
    // epilogue: acc[t][r] = (centroid row cl(t, r, half), query column l31)
    float gm[2] = {-INFINITY, -INFINITY};
    const int nq_w = nq - qw;  // queries of this wave that exist (may be <= 0)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = (r & 3) + 8 * (r >> 2) + 4 * half;   // row inside the tile
        float v = 2.f * acc[t][r];
        v = v - q2;
        v = v - c2s[t * 32 + cl];
        gm[t >> 2] = fmaxf(gm[t >> 2], v);
        trw[l31 * 33 + cl] = v;                              // [query][centroid]
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      // read back transposed: lane (l31, half) takes centroid l31 of queries 16 half + i
      const int c = c0 + t * 32 + l31;
      if (c < n_cells) {
        float* __restrict__ out = sims + (int64_t)(qw + 16 * half) * n_cells + c;
        const int n_here = nq_w - 16 * half;  // rows of this half-wave that exist
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = trw[(16 * half + i) * 33 + l31];
          if (i < n_here) out[(int64_t)i * n_cells] = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_sched_barrier(0);  // one tile at a time: keeps the live predicate masks few
    }
    // the two half-waves of a query hold disjoint centroid rows
    gm[0] = fmaxf(gm[0], __shfl_xor(gm[0], 32, 64));
    gm[1] = fmaxf(gm[1], __shfl_xor(gm[1], 32, 64));
    if (half == 0 && qvalid) {
      const int g0 = 2 * ch;
      gmax[(int64_t)q * n_groups + g0] = gm[0];
      if (g0 + 1 < n_groups) gmax[(int64_t)q * n_groups + g0 + 1] = gm[1];
    }
  }
}

// Small problems (few centroid groups x query chunks): 64-query x (64 CT)-centroid tiles, both
// operands through LDS in double-buffered k-batches of 16.  CT = 4 (256 centroids per block): twice
// the blocks of the kernel above; CT = 1 (64 centroids): eight times -- a 1000-query GIST batch
// (d = 960, 1024 cells) is 64 blocks at CT = 4, a quarter of the chip each walking 960 dimensions,
// and 256 at CT = 1.
constexpr int kCsKB = 16;

template <int CT>
__global__ __launch_bounds__(256) void coarse_sims_small_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ C,
                                                         float* __restrict__ sims, int d, int nq,
                                                         int n_cells) {
  constexpr int W = 64 * CT;  // centroids per block: 2 wave columns x CT tiles x 32
  __shared__ float As[2][kCsKB][64];
  __shared__ float Bs[2][kCsKB][W];
  __shared__ float q2s[64];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int qb = blockIdx.x * 64, cb = blockIdx.y * W;
  const int wq = 32 * (wave & 1), wc = 32 * CT * (wave >> 1);

  // staging: A batch = 16 rows x 64 queries (4 elements per thread), B batch = 16 rows x W
  // centroids (4 CT per thread); a thread's elements of one row are contiguous across the wave
  const int a_col = tid & 63, a_row0 = tid >> 6;  // rows a_row0 + 4u
  const bool a_ok = qb + a_col < nq;
  constexpr int BR = 256 / W;                     // B rows covered by one pass of the block (1 or 4)
  const int b_col = tid % W, b_row0 = tid / W;    // rows b_row0 + BR u
  const bool b_ok = cb + b_col < n_cells;
  const float* __restrict__ xa = x + (a_ok ? qb + a_col : 0);
  const float* __restrict__ cbp = C + (b_ok ? cb + b_col : 0);
  float ra[4], rb[kCsKB / BR];
  auto load_batch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + a_row0 + 4 * u;
      ra[u] = (a_ok && k < d) ? xa[(int64_t)k * nq] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kCsKB / BR; ++u) {
      const int k = k0 + b_row0 + BR * u;
      rb[u] = (b_ok && k < d) ? cbp[(int64_t)k * n_cells] : 0.f;
    }
  };
  auto store_batch = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) As[buf][a_row0 + 4 * u][a_col] = ra[u];
#pragma unroll
    for (int u = 0; u < kCsKB / BR; ++u) Bs[buf][b_row0 + BR * u][b_col] = rb[u];
  };

  f32x16 acc[CT];
  float b2[CT], a2 = 0.f;
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    b2[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  }
  const int n_batches = (d + kCsKB - 1) / kCsKB;
  load_batch(0);
  store_batch(0);
  __syncthreads();
  for (int bt = 0; bt < n_batches; ++bt) {
    const int buf = bt & 1;
    if (bt + 1 < n_batches) load_batch((bt + 1) * kCsKB);
#pragma unroll
    for (int kk = 0; kk < kCsKB / 2; ++kk) {
      // both k rows of the step in every lane: the norms are ONE ascending-k fmaf chain, the same
      // arithmetic as coarse_sims_kernel (and oracle_coarse_sims) -- a sim does not depend on
      // which of the kernels the batch size selects.  (Even / odd partial chains added at the end
      // differed from it in the last bit.)
      const float a0 = As[buf][2 * kk][wq + l31], a1 = As[buf][2 * kk + 1][wq + l31];
      a2 = fmaf(a0, a0, a2);
      a2 = fmaf(a1, a1, a2);
      const float a = half ? a1 : a0;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const float b0 = Bs[buf][2 * kk][wc + 32 * t + l31], b1 = Bs[buf][2 * kk + 1][wc + 32 * t + l31];
        b2[t] = fmaf(b0, b0, b2[t]);
        b2[t] = fmaf(b1, b1, b2[t]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, half ? b1 : b0, acc[t], 0, 0, 0);
      }
    }
    if (bt + 1 < n_batches) store_batch(buf ^ 1);
    __syncthreads();
  }
  if (wave < 2 && half == 0) q2s[32 * wave + l31] = a2;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int c = cb + wc + 32 * t + l31;
    if (c >= n_cells) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int qq = qb + wq + row;
      if (qq < nq) {
        float v = 2.f * acc[t][r];
        v = v - q2s[wq + row];
        v = v - b2[t];
        sims[(int64_t)qq * n_cells + c] = v;
      }
    }
  }
}

template <int R>
static int launch_select(const float* x, const float* a2, const float* b2, float* v, int64_t* i,
                         int rows, int cols, int k, hipStream_t st, const ProbeEpilogue& pe,
                         const GroupFilter& gf) {
  hipLaunchKernelGGL(topk_select_kernel<R>, dim3((rows + kSelWaves - 1) / kSelWaves),
                     dim3(kSelWaves * 64), 0, st, x, a2, b2, v, i, rows, cols, k, pe, gf);
  TPQ_LAUNCH_CHECK("topk_select_kernel");
  return TPQ_OK;
}

}  // namespace tpq

using namespace tpq;

static int select_impl(const float* x, const float* a2, const float* b2, float* vals, int64_t* idx,
                       int rows, int cols, int k, tpq_stream_t stream,
                       const ProbeEpilogue& pe = ProbeEpilogue{}, const GroupFilter& gf = GroupFilter{});

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
                       int rows, int cols, int k, tpq_stream_t stream, const ProbeEpilogue& pe,
                       const GroupFilter& gf) {
  TPQ_REQUIRE(x && vals && idx, "topk_select: null pointer");
  TPQ_REQUIRE(rows >= 0 && cols >= 1, "topk_select: bad shape [%d, %d]", rows, cols);
  TPQ_REQUIRE(k >= 1 && k <= 1024 && k <= cols, "topk_select: k=%d out of range (cols=%d, max 1024)", k, cols);
  if (rows == 0) return TPQ_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int r = (k + 63) / 64;
  if (r <= 1) return launch_select<1>(x, a2, b2, vals, idx, rows, cols, k, st, pe, gf);
  if (r <= 2) return launch_select<2>(x, a2, b2, vals, idx, rows, cols, k, st, pe, gf);
  if (r <= 4) return launch_select<4>(x, a2, b2, vals, idx, rows, cols, k, st, pe, gf);
  if (r <= 8) return launch_select<8>(x, a2, b2, vals, idx, rows, cols, k, st, pe, gf);
  return launch_select<16>(x, a2, b2, vals, idx, rows, cols, k, st, pe, gf);
}

// Which arithmetic selects (results are the same either way, bit for bit):
//   TPQ_PROBE_ROUTE_AUTO   the fp16 selection pass + exact candidates from kProbeFastMinCells cells on (1 024 for large batches, few probes), where the fp32
//                          similarity GEMM dominates the coarse step; the fp32 kernels below
//   TPQ_PROBE_ROUTE_FP32   the fp32-MFMA similarity kernels always
//   TPQ_PROBE_ROUTE_FP16   the fp16 selection pass whenever the shape supports it (use_tensor_core=True)
constexpr int kProbeFastMinCells = 2048;
static bool probe_fast_route(int d, int nq, int n_cells, int n_probe, int route);

static size_t probe_fp32_workspace_bytes(int nq, int n_cells) {
  // sims [nq][n_cells] + group maxima [nq][ceil(n_cells / 128)]
  return ((size_t)nq * (size_t)n_cells + (size_t)nq * (size_t)((n_cells + 127) / 128)) * sizeof(float);
}
extern "C" size_t tpq_ivfpq_coarse_probe_route_workspace_bytes(int d, int nq, int n_cells, int route) {
  if (nq <= 0 || n_cells <= 0) return 0;
  const size_t plain = probe_fp32_workspace_bytes(nq, n_cells);
  if (route == TPQ_PROBE_ROUTE_FP32 || !lloyd_probe_supported(d, nq, n_cells)) return plain;
  const size_t fast = lloyd_probe_workspace_bytes(d, nq, n_cells);
  return fast > plain ? fast : plain;
}
extern "C" size_t tpq_ivfpq_coarse_probe_workspace_bytes(int nq, int n_cells) {
  if (nq <= 0 || n_cells <= 0) return 0;
  return tpq_ivfpq_coarse_probe_route_workspace_bytes(128, nq, n_cells, TPQ_PROBE_ROUTE_AUTO);  // (covers every d <= 128)
}

extern "C" size_t tpq_ivfpq_coarse_probe_prepared_bytes(int d, int n_cells) {
  return lloyd_probe_prepared_bytes(d, n_cells);
}
extern "C" int tpq_ivfpq_coarse_probe_prepare(const float* centroids, int d, int n_cells, void* prepared,
                                              size_t prepared_bytes, tpq_stream_t stream) {
  TPQ_REQUIRE(centroids && prepared, "ivfpq_coarse_probe_prepare: null pointer");
  const size_t need = lloyd_probe_prepared_bytes(d, n_cells);
  if (need == 0) {
    set_error("ivfpq_coarse_probe_prepare: shape d=%d n_cells=%d has no fp16 selection pass (d <= 128, n_cells %% 32 == 0)",
              d, n_cells);
    return TPQ_ERR_UNSUPPORTED;
  }
  TPQ_REQUIRE(prepared_bytes >= need, "ivfpq_coarse_probe_prepare: prepared block of %zu bytes needed", need);
  return lloyd_probe_prepare(centroids, d, n_cells, reinterpret_cast<char*>(prepared), reinterpret_cast<hipStream_t>(stream));
}

static bool probe_fast_route(int d, int nq, int n_cells, int n_probe, int route) {
  if (route == TPQ_PROBE_ROUTE_FP32 || !lloyd_probe_supported(d, nq, n_cells)) return false;
  if (n_probe + 16 > 1024) return false;  // (the candidate list: 64 R >= n_probe + 16 entries, R <= 16)
  if (route == TPQ_PROBE_ROUTE_FP16) return true;
  // (beyond 112 probes the candidate list takes four registers per lane and the fast select's folds cost more than
  // the fp32 GEMM saves: 16 384 cells, 128 probes: 1.15 ms against 0.83; 64 probes: 0.38 against 0.67)
  // ... unless the direct candidate list applies (2 n_probe <= groups of cells: 16 384 cells in 256 groups, 128 probes:
  // 0.41 ms against 0.77)
  if (n_cells >= kProbeFastMinCells)
    return nq > kProbeSmallMaxQ && (n_probe <= 112 || 2 * n_probe <= lloyd_probe_groups(n_cells));
  // (1 024 cells, 10 000 queries: 0.054-0.079 ms against 0.082-0.090 up to 32 probes; 1 000 queries: 0.035 against 0.025)
  return n_cells >= 1024 && nq >= 4096 && n_probe <= 32;
}

template <int R>
static int launch_probe_fast(const ProbeFastBuffers& fb, const float* x, float* vals, int64_t* idx, int d, int nq,
                             int n_cells, int k, const ProbeEpilogue& pe, hipStream_t st) {
  hipLaunchKernelGGL(probe_select_fast_kernel<R>, dim3((nq + kSelWaves - 1) / kSelWaves), dim3(kSelWaves * 64), 0, st,
                     fb, x, vals, idx, d, nq, n_cells, k, pe);
  TPQ_LAUNCH_CHECK("probe_select_fast_kernel");
  return TPQ_OK;
}

extern "C" int tpq_ivfpq_coarse_probe(const float* query, const float* centroids,
                                      const int64_t* cell_start_tbl, const int64_t* cell_size_tbl,
                                      float* topk_sims, int64_t* cells, int64_t* cell_start,
                                      int64_t* cell_size, int64_t* n_probe_list, int d, int nq,
                                      int n_cells, int n_probe, float smart_temperature,
                                      void* workspace, size_t workspace_bytes,
                                      tpq_stream_t stream) {
  return tpq_ivfpq_coarse_probe_route(query, centroids, cell_start_tbl, cell_size_tbl, topk_sims, cells, cell_start,
                                      cell_size, n_probe_list, d, nq, n_cells, n_probe, smart_temperature,
                                      TPQ_PROBE_ROUTE_AUTO, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int tpq_ivfpq_coarse_probe_route(const float* query, const float* centroids,
                                            const int64_t* cell_start_tbl, const int64_t* cell_size_tbl,
                                            float* topk_sims, int64_t* cells, int64_t* cell_start,
                                            int64_t* cell_size, int64_t* n_probe_list, int d, int nq,
                                            int n_cells, int n_probe, float smart_temperature, int route,
                                            const void* prepared, void* workspace, size_t workspace_bytes,
                                            tpq_stream_t stream) {
  TPQ_REQUIRE(route == TPQ_PROBE_ROUTE_AUTO || route == TPQ_PROBE_ROUTE_FP32 || route == TPQ_PROBE_ROUTE_FP16,
              "ivfpq_coarse_probe: bad route %d", route);
  TPQ_REQUIRE(query && centroids && cell_start_tbl && cell_size_tbl && topk_sims && cells &&
                  cell_start && cell_size && n_probe_list,
              "ivfpq_coarse_probe: null pointer argument");
  TPQ_REQUIRE(d >= 1 && nq >= 0 && n_cells >= 1, "ivfpq_coarse_probe: bad shape d=%d nq=%d n_cells=%d",
              d, nq, n_cells);
  TPQ_REQUIRE(n_probe >= 1 && n_probe <= n_cells && n_probe <= 1024,
              "ivfpq_coarse_probe: n_probe=%d out of range (n_cells=%d, max 1024)", n_probe, n_cells);
  if (nq == 0) return TPQ_OK;
  const size_t need = tpq_ivfpq_coarse_probe_route_workspace_bytes(d, nq, n_cells, route);
  if (!workspace || workspace_bytes < need) {
    set_error("ivfpq_coarse_probe: workspace too small (%zu < %zu)", workspace_bytes, need);
    return TPQ_ERR_WORKSPACE;
  }
  float* sims = reinterpret_cast<float*>(workspace);
  ProbeEpilogue pe{cell_start_tbl, cell_size_tbl, cell_start, cell_size, n_probe_list,
                   smart_temperature > 0.f ? 1.0f / smart_temperature : 0.f};
  if (probe_fast_route(d, nq, n_cells, n_probe, route)) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ProbeFastBuffers fb;
    int rc = lloyd_probe_sims(query, centroids, prepared, d, nq, n_cells, reinterpret_cast<char*>(workspace), &fb, st);
    if (rc) return rc;
    const int r = (n_probe + 16 + 63) / 64;
    if (r <= 1) return launch_probe_fast<1>(fb, query, topk_sims, cells, d, nq, n_cells, n_probe, pe, st);
    if (r <= 2) return launch_probe_fast<2>(fb, query, topk_sims, cells, d, nq, n_cells, n_probe, pe, st);
    if (r <= 4) return launch_probe_fast<4>(fb, query, topk_sims, cells, d, nq, n_cells, n_probe, pe, st);
    if (r <= 8) return launch_probe_fast<8>(fb, query, topk_sims, cells, d, nq, n_cells, n_probe, pe, st);
    return launch_probe_fast<16>(fb, query, topk_sims, cells, d, nq, n_cells, n_probe, pe, st);
  }
  if (nq <= kProbeSmallMaxQ && n_cells <= kProbeSmallMaxCells && d <= kProbeSmallMaxD &&
      (long long)n_cells * d <= (1 << 20)) {  // one launch: sims row in LDS + select, one block per query
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int r = (n_probe + 63) / 64;
#define TPQ_PS(RR)                                                                                         \
  hipLaunchKernelGGL(probe_small_kernel<RR>, dim3(nq), dim3(kProbeSmallThreads), 0, st, query, centroids, \
                     topk_sims, cells, d, nq, n_cells, n_probe, pe)
    if (r <= 1) TPQ_PS(1);
    else if (r <= 2) TPQ_PS(2);
    else if (r <= 4) TPQ_PS(4);
    else if (r <= 8) TPQ_PS(8);
    else TPQ_PS(16);
#undef TPQ_PS
    TPQ_LAUNCH_CHECK("probe_small_kernel");
    return TPQ_OK;
  }
  // large problems: blocks = 128-query groups x centroid-chunk groups (a block walks several
  // 256-centroid chunks once there are enough blocks to fill the chip a few times over) and the
  // row select is restricted by the group maxima; small ones: 64 x 256 tiles, full row select
  const int qgroups = (nq + 127) / 128, chunks = (n_cells + kCsRows - 1) / kCsRows;
  const int n_groups = (n_cells + 127) / 128;
  float* gmax = sims + (size_t)nq * n_cells;
  GroupFilter gf{nullptr, 0};
  if ((long long)qgroups * chunks < 512) {
    const long long blocks4 = (long long)((nq + 63) / 64) * ((n_cells + 255) / 256);
    if (blocks4 < 192)  // under three quarters of the CUs: 64-centroid tiles, 4x the blocks
      hipLaunchKernelGGL(coarse_sims_small_kernel<1>, dim3((nq + 63) / 64, (n_cells + 63) / 64),
                         dim3(256), 0, reinterpret_cast<hipStream_t>(stream), query, centroids, sims,
                         d, nq, n_cells);
    else
      hipLaunchKernelGGL(coarse_sims_small_kernel<4>, dim3((nq + 63) / 64, (n_cells + 255) / 256),
                         dim3(256), 0, reinterpret_cast<hipStream_t>(stream), query, centroids, sims,
                         d, nq, n_cells);
  } else {
    int per_block = (int)(((long long)qgroups * chunks) / 1024);
    per_block = per_block < 1 ? 1 : (per_block > 8 ? 8 : per_block);
    hipLaunchKernelGGL(coarse_sims_kernel, dim3(qgroups, (chunks + per_block - 1) / per_block),
                       dim3(256), 0, reinterpret_cast<hipStream_t>(stream), query, centroids, sims,
                       d, nq, n_cells, per_block, gmax, n_groups);
    gf = GroupFilter{gmax, n_groups};
  }
  TPQ_LAUNCH_CHECK("coarse_sims_kernel");
  return select_impl(sims, nullptr, nullptr, topk_sims, cells, nq, n_cells, n_probe, stream, pe, gf);
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
