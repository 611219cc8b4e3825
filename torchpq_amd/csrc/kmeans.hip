// K-means assign (batched pairwise similarity + arg-max) and update for gfx950.
//
// tpq_max_sim replaces MaxSimCuda(A, B, dim=2, mode="tn") (torchpq/kernels/MaxSimCuda.py:184-238,
// kernel max_sim_tn torchpq/kernels/cuda/max_sim.cu:182-309): the reference is a CUDA-core
// 128x128 SGEMM-like tile with a cross-block float atomicMax + racy index store (:152-180).
// Here the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32, ascending-k fma chain):
// centroids are the MFMA rows, data points the MFMA columns, so each lane owns ONE point and
// the arg-max over centroids is an in-lane reduction over accumulator registers -- a block
// sees every centroid for its 128 points, so there is no cross-block reduction and no race.
//
// tpq_compute_centroids replaces compute_centroids (torchpq/kernels/cuda/compute_centroids.cu:10-86):
// the reference launches l*d blocks that each re-read all labels; here data and labels are read
// exactly once (LDS atomics per block, one global atomic flush, tiny finalize kernel).
#include <type_traits>

#include "common.h"

namespace tpq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(integral_constant<int, I>) for I in [I0, I1)
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

// (best, besti) <- (val, CL) if val > best: compare + two selects.  CL must be an inline constant
// (0..64): a 32-bit literal next to vcc violates the one-constant-bus-operand rule of VOP2, which is
// why the compiler spends a v_mov per candidate index; the callers therefore track the index
// within the 32-centroid unit (0..27) here and the unit number once per unit.
// volatile: pins the selects next to the compare (left alone, the optimiser sinks the besti chain
// to its use at the end of the tile and parks 128 compare masks in spilled SGPRs)
template <int CL>
__device__ __forceinline__ void ms_take(float& best, int& besti, float val) {
  static_assert(CL >= 0 && CL <= 64, "inline constant");
  asm volatile(
      "v_cmp_ngt_f32 vcc, %2, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, %2, %0, vcc\n\tv_cndmask_b32 %1, %3, %1, vcc"
      : "+v"(best), "+v"(besti)
      : "v"(val), "n"(CL)
      : "vcc");
}

constexpr int kMsCent = 256;  // centroids per pass (8 MFMA row tiles)
constexpr int kMsKC = 16;     // k rows staged in LDS per step (per buffer)

// grid (ceil(m/128), l), block 256 = 4 waves x 32 points.  Centroid chunk (256 rows) x k-slab (16)
// tiles are double-buffered in LDS: the next slab goes global -> registers while the MFMAs of
// the current one run, then registers -> the other buffer, one barrier per slab; the wave's own
// point operand for the next slab is prefetched the same way.  (The first version staged and
// consumed each slab between two barriers and loaded its point operand inside the MFMA loop:
// 17-23 TF/s.)
constexpr int kMsSlab = kMsKC * kMsCent;  // floats per LDS buffer (16 KiB)

__global__ __launch_bounds__(256, 2) void max_sim_kernel(const float* __restrict__ A,
                                                         const float* __restrict__ B,
                                                         float* __restrict__ vals,
                                                         int64_t* __restrict__ inds, int d, int m,
                                                         int n, int euclidean,
                                                         const int* __restrict__ list,
                                                         const int* __restrict__ count,
                                                         unsigned long long* __restrict__ keys,
                                                         const float* __restrict__ Ac, int cap, int pos0,
                                                         int tile_stride) {
  // list != nullptr (tpq_coarse_assign's exact re-check): the points are columns list[0 .. *count)
  // of A, results go to inds[list[p]], vals may be null; the grid covers the worst case and blocks
  // beyond *count leave at once.  keys != nullptr (one problem, many centroids, few points): the
  // CENTROIDS are split over gridDim.y blocks per point tile -- a few hundred listed points would
  // otherwise occupy a few hundred long-running blocks, one per CU -- and every split folds its
  // (value, index) into keys[point] with a 64-bit atomicMax (order-preserving value bits, then
  // ~index: ties go to the smaller index); max_sim_list_decode_kernel writes the results.
  // Ac != nullptr: list positions [0, min(*count, cap)) read their point from the compact copy
  // Ac [d][cap] (gather_columns_kernel) -- coalesced; gathering the listed columns of A inside the
  // slab loop (64 cache lines per load instruction, once per centroid chunk) was what the re-check
  // spent its time on.  A second launch (Ac == nullptr, pos0 = cap) gathers for the overflow, if any.
  extern __shared__ __attribute__((aligned(16))) float ms_smem[];
  float* cs = ms_smem;                  // [2][kMsKC][kMsCent]
  float* b2s = ms_smem + 2 * kMsSlab;   // [kMsCent]
  const int b = keys ? 0 : blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  int m_eff = m;
  if (list) {
    list += (int64_t)b * m;  // one list per sub-problem
    m_eff = count[b];
    m_eff = m_eff < m ? m_eff : m;
    if (Ac) m_eff = m_eff < cap ? m_eff : cap;
  }
  // tile_stride > 0 (batched lists, tpq_lloyd_step): the grid is a FEW blocks per sub-problem, each
  // walking the 128-point tiles tile0, tile0 + tile_stride, ... of its list -- a grid over the worst
  // case (every point listed) is 500 000 blocks at configs[4], nearly all of them empty, and costs
  // more than the listed 0.3 % of the points do
  for (int tile0 = blockIdx.x;; tile0 += tile_stride) {
  const int pos = pos0 + tile0 * 128 + wave * 32 + l31;  // this lane's point (position in the list)
  if (list && pos0 + tile0 * 128 >= m_eff) return;  // block-uniform
  if (!list && tile0 * 128 >= m) return;
  const bool iv = pos < m_eff;
  const int i = list ? (iv ? list[pos] : 0) : pos;  // column of A / slot of the outputs
  // Every global load below is `uniform row pointer [per-lane 32-bit offset]`: the row pointer
  // lives in SGPRs, the offset in ONE VGPR per operand, rows past d are clamped to d-1 and
  // neutralised afterwards (no exec-mask branches).  (Round 1 formed a 64-bit per-lane address
  // for each of the 24 loads of a slab: 48 address registers, the kernel sat at its 256-VGPR cap
  // with two registers left for the MFMA A operands, and every pair of MFMAs waited for its own
  // LDS round trip -- ds_read2, s_waitcnt lgkmcnt(0), 2 MFMAs, repeat: 80-86 TF/s.)
  const float* __restrict__ Ab = Ac ? Ac : A + (int64_t)b * d * m;  // uniform
  const float* __restrict__ Bb = B + (int64_t)b * d * n;  // uniform
  const int a_cols = Ac ? cap : m;
  const uint32_t xoff = iv ? (uint32_t)(Ac ? pos : i) : 0u;  // this lane's column of A (a valid one)
  auto a_row = [&](int k) -> const float* { return Ab + (int64_t)(k < d ? k : d - 1) * a_cols; };
  auto b_row = [&](int k) -> const float* { return Bb + (int64_t)(k < d ? k : d - 1) * n; };

  // |a|^2, one ascending-k fma chain per point -- formed from the MFMA operands of the FIRST centroid chunk's
  // slabs (a lane holds the dimensions of its parity; one half-swap per pair hands every lane both), not by a
  // pass of its own: that pass read every point's column a second and (both half-waves) a third time, and for
  // listed points every one of those reads is a cache line of its own -- level 3 of the Lloyd step (0.3 % of
  // 64 M points scattered over 16 GB) spent 0.55 ms mostly there.
  float a2 = 0.f;

  float best = -INFINITY;
  int besti = 0;
  const int n_slabs = (d + kMsKC - 1) / kMsKC;

  int c_lo = 0, c_hi = n;
  if (keys) {  // this block's share of the centroid chunks
    const int chunks = (n + kMsCent - 1) / kMsCent;
    c_lo = (int)(((int64_t)chunks * blockIdx.y) / gridDim.y) * kMsCent;
    c_hi = (int)(((int64_t)chunks * (blockIdx.y + 1)) / gridDim.y) * kMsCent;
    c_hi = c_hi < n ? c_hi : n;
  }
  for (int c0 = c_lo; c0 < c_hi; c0 += kMsCent) {
    const int nc = (c_hi - c0) < kMsCent ? (c_hi - c0) : kMsCent;
    const int nt = (nc + 31) >> 5;
    const bool cv = (int)threadIdx.x < nc;  // this thread's centroid column of the chunk exists
    const uint32_t coff = (uint32_t)c0 + (cv ? threadIdx.x : 0u);
    float rs[kMsKC];       // staged slab: row u, column threadIdx.x
    float xc[kMsKC / 2], xn[kMsKC / 2];
    auto load_slab = [&](int kb) {
#pragma unroll
      for (int u = 0; u < kMsKC; ++u) rs[u] = b_row(kb + u)[coff];
    };
    // rows past d must multiply as zero: the centroid side is zeroed (uniform condition), so the
    // point side may hold a clamped row
    auto mask_slab = [&](int kb) {
#pragma unroll
      for (int u = 0; u < kMsKC; ++u) rs[u] = (kb + u < d) ? rs[u] : 0.f;
    };
    auto store_slab = [&](float* dst) {
#pragma unroll
      for (int u = 0; u < kMsKC; ++u) dst[u * kMsCent + threadIdx.x] = rs[u];
    };
    auto load_x = [&](int kb, float (&x)[kMsKC / 2]) {
      // B operand [k][col=point]: this half-wave owns k = kb + 2j + half
#pragma unroll
      for (int j = 0; j < kMsKC / 2; ++j) {
        const float* r0 = a_row(kb + 2 * j);
        const float* r1 = a_row(kb + 2 * j + 1);
        x[j] = (half ? r1 : r0)[xoff];
      }
    };
    // |b|^2 of the chunk's centroids comes for free: thread t stages column t of every slab, in
    // ascending k, so the squares of what it stages form the ascending-k fma chain of centroid t
    float bsq = 0.f;
    auto square_slab = [&]() {
#pragma unroll
      for (int u = 0; u < kMsKC; ++u) bsq = fmaf(rs[u], rs[u], bsq);
    };
    load_slab(0);
    load_x(0, xc);
    __syncthreads();  // every wave finished the previous chunk (reads of cs and b2s)
    mask_slab(0);
    square_slab();
    if (n_slabs == 1) b2s[threadIdx.x] = bsq;
    store_slab(cs);
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();

    for (int sb = 0; sb < n_slabs; ++sb) {
      const float* cur = cs + (sb & 1) * kMsSlab;
      const bool more = sb + 1 < n_slabs;
      if (more) {
        load_slab((sb + 1) * kMsKC);
        load_x((sb + 1) * kMsKC, xn);
      }
      // (rows beyond the chunk's last centroid are staged as zeros and masked in the epilogue, so
      // all 8 row tiles are always multiplied: no per-MFMA predicate in the hot loop)
      // A operands (8 centroid rows per k-step) are read from LDS one k-step AHEAD of their MFMAs
      const float* crow0 = cur + half * kMsCent + l31;  // A operand [row=centroid][k = 2j + half]
      float an[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) an[t] = crow0[t * 32];
#pragma unroll
      for (int j = 0; j < kMsKC / 2; ++j) {
        float ac[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) ac[t] = an[t];
        if (j + 1 < kMsKC / 2) {
#pragma unroll
          for (int t = 0; t < 8; ++t) an[t] = crow0[(2 * (j + 1)) * kMsCent + t * 32];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t], xc[j], acc[t], 0, 0, 0);
        if (euclidean && c0 == c_lo) {  // (block-uniform) the two squares of this pair of dimensions, ascending
          float xe = xc[j], xo = xc[j];
          // (inline asm, hazard wait states inside the string: max_sim_codebook_kernel's note)
          asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xe), "+v"(xo));
          const int k = sb * kMsKC + 2 * j;  // rows past d hold a clamped row: they are not part of |a|^2
          xe = k < d ? xe : 0.f;
          xo = k + 1 < d ? xo : 0.f;
          a2 = fmaf(xe, xe, a2);
          a2 = fmaf(xo, xo, a2);
        }
        // order inside the k-step: the next k-step's LDS reads first, then the 8 MFMAs (left to
        // itself the scheduler sinks the reads behind six of the MFMAs and the next k-step starts
        // by waiting for them); the barrier keeps later k-steps' reads from being hoisted here
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // MFMA
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) {
        mask_slab((sb + 1) * kMsKC);
        square_slab();
        if (sb + 2 == n_slabs) b2s[threadIdx.x] = bsq;  // the chain is complete
        store_slab(cs + ((sb + 1) & 1) * kMsSlab);
#pragma unroll
        for (int j = 0; j < kMsKC / 2; ++j) xc[j] = xn[j];
      }
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (cl < nc) {
            float v = acc[t][r];
            if (euclidean) {
              v = 2.f * v;
              v = v - a2;
              v = v - b2s[cl];
            }
            const int c = c0 + cl;
            if (v > best || (v == best && c < besti)) {
              best = v;
              besti = c;
            }
          }
        }
      }
    }
  }
  // the two half-waves hold disjoint centroid rows of the same point
  const float ov = __shfl_xor(best, 32, 64);
  const int oi = __shfl_xor(besti, 32, 64);
  if (ov > best || (ov == best && oi < besti)) {
    best = ov;
    besti = oi;
  }
  if (half == 0 && iv) {
    if (keys) {
      if (c_lo < c_hi) {
        const unsigned fb = __float_as_uint(best);
        const unsigned ordered = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);
        atomicMax(keys + i, ((unsigned long long)ordered << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)besti));
      }
    } else {
      if (vals) vals[(int64_t)b * m + i] = best;
      inds[(int64_t)b * m + i] = besti;
    }
  }
  if (tile_stride <= 0) return;
  __syncthreads();  // the next tile's first slab overwrites cs / b2s
  }
}

// Ac[k][p] = A[k][list[p]] for p < min(*count, cap): grid (ceil(cap / 256), d)
__global__ __launch_bounds__(256) void gather_columns_kernel(const float* __restrict__ A,
                                                            const int* __restrict__ list,
                                                            const int* __restrict__ count,
                                                            float* __restrict__ Ac, int m, int cap) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int cnt = *count < cap ? *count : cap;
  if (p >= cnt) return;
  Ac[(int64_t)blockIdx.y * cap + p] = A[(int64_t)blockIdx.y * m + list[p]];
}

__global__ __launch_bounds__(256) void max_sim_list_decode_kernel(const int* __restrict__ list,
                                                                 const int* __restrict__ count,
                                                                 const unsigned long long* __restrict__ keys,
                                                                 float* __restrict__ vals,
                                                                 int64_t* __restrict__ inds, int m) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int cnt = *count < m ? *count : m;
  if (p >= cnt) return;
  const int i = list[p];
  const unsigned long long key = keys[i];
  const unsigned ordered = (unsigned)(key >> 32);
  const unsigned fb = (ordered & 0x80000000u) ? (ordered & 0x7FFFFFFFu) : ~ordered;
  inds[i] = (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  if (vals) vals[i] = __uint_as_float(fb);
}

// the exact kernel over device-side lists of points (list [l][m], count [l]): see tpq_coarse_assign /
// tpq_max_sim_select
int launch_max_sim_list(const float* A, const float* B, float* vals, int64_t* inds, int l, int d, int m, int n,
                        int euclid, const int* list, const int* count, unsigned long long* keys, float* Ac, int cap,
                        hipStream_t st) {
  const size_t ms_lds = (size_t)(2 * kMsSlab + kMsCent) * sizeof(float);
  int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(max_sim_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ms_lds),
                     "max_sim_kernel attr");
  if (rc) return rc;
  if (keys && l == 1) {
    // one problem: keys [m] u64 (zeroed by the caller) + compact copy Ac [d][cap]; the centroid
    // chunks are split up to 8 ways
    const int chunks = (n + kMsCent - 1) / kMsCent;
    const int splits = chunks < 8 ? chunks : 8;
    cap = cap < m ? cap : m;
    hipLaunchKernelGGL(gather_columns_kernel, dim3((cap + 255) / 256, d), dim3(256), 0, st, A, list, count, Ac, m,
                       cap);
    TPQ_LAUNCH_CHECK("gather_columns_kernel");
    hipLaunchKernelGGL(max_sim_kernel, dim3((cap + 127) / 128, splits), dim3(256), ms_lds, st, A, B, vals, inds, d,
                       m, n, euclid, list, count, keys, static_cast<const float*>(Ac), cap, 0, 0);
    TPQ_LAUNCH_CHECK("max_sim_kernel (list, compact)");
    if (cap < m) {  // more listed points than the compact copy holds: gather for the rest
      hipLaunchKernelGGL(max_sim_kernel, dim3((m - cap + 127) / 128, splits), dim3(256), ms_lds, st, A, B, vals,
                         inds, d, m, n, euclid, list, count, keys, static_cast<const float*>(nullptr), 0, cap, 0);
      TPQ_LAUNCH_CHECK("max_sim_kernel (list, overflow)");
    }
    hipLaunchKernelGGL(max_sim_list_decode_kernel, dim3((m + 255) / 256), dim3(256), 0, st, list, count, keys,
                       vals, inds, m);
    TPQ_LAUNCH_CHECK("max_sim_list_decode_kernel");
    return TPQ_OK;
  }
  // a few looping blocks per sub-problem (see the kernel): all CUs busy twice over, no empty blocks
  int per = (2048 + l - 1) / l;
  per = per < (m + 127) / 128 ? per : (m + 127) / 128;
  hipLaunchKernelGGL(max_sim_kernel, dim3(per, l), dim3(256), ms_lds, st, A, B, vals, inds, d,
                     m, n, euclid, list, count, static_cast<unsigned long long*>(nullptr),
                     static_cast<const float*>(nullptr), 0, 0, per);
  TPQ_LAUNCH_CHECK("max_sim_kernel (list)");
  return TPQ_OK;
}

// ---- assign, codebook-sized problems (n <= 256 centroids, d <= 128): the PQ train/encode shape --
// All centroids of sub-problem b are staged in LDS ONCE per block ([d][256] fp32, <= 128 KiB) and
// the block then walks kMsTiles point tiles.  Per tile a wave pre-loads its whole data fragment
// (DH = d/2 registers, loads issued back to back) before the MFMA loop, so the matrix pipe is fed
// from registers + LDS only: 8 independent 32x32 accumulators, 8 ds_read_b32 per 8 MFMAs.
#ifndef TPQ_MS_TILES
#define TPQ_MS_TILES 32
#endif
constexpr int kMsTiles = TPQ_MS_TILES;  // 128-point tiles per block

template <int DH, bool euclidean>
__global__ __launch_bounds__(256, (DH <= 32 ? 2 : 1)) void max_sim_codebook_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ vals,
    int64_t* __restrict__ inds, int d, int m, int n_total, int c0, int first) {
  // this launch covers centroids [c0, c0 + n) of the n_total; later chunks fold their best into
  // the running (vals, inds) -- ascending chunks, so on a tie the earlier (smaller) index stays
  const int n = (n_total - c0) < 256 ? (n_total - c0) : 256;
  extern __shared__ __attribute__((aligned(16))) float msh[];
  float* cs = msh;                   // [2*DH][256], zero beyond (d, n)
  float* b2s = msh + 2 * DH * 256;   // [256]: |b|^2 (euclidean) or 0 (inner); padding columns
                                     // c >= n hold +inf / -inf so that they can never win
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const float* __restrict__ Ab = A + (int64_t)b * d * m;
  const float* __restrict__ Bb = B + (int64_t)b * d * n_total + c0;

  for (int e = threadIdx.x; e < 2 * DH * 256; e += 256) {
    const int k = e >> 8, c = e & 255;
    cs[e] = (k < d && c < n) ? Bb[(int64_t)k * n_total + c] : 0.f;
  }
  __syncthreads();
  {
    const int c = threadIdx.x;
    float s = 0.f;
    if (euclidean) {
      for (int k = 0; k < d; ++k) s = fmaf(cs[k * 256 + c], cs[k * 256 + c], s);
      if (c >= n) s = INFINITY;
    } else if (c >= n) {
      s = -INFINITY;
    }
    b2s[c] = s;
  }
  __syncthreads();

  // software pipeline over the block's tiles: the data fragment of tile t+1 is in flight while
  // tile t runs its 256 MFMAs
  // SRSRC buffer loads: 32-bit per-lane offset + uniform row offset in an SGPR, so the 2*DH loads
  // of a fragment need no per-load 64-bit address registers (what made a prefetched fragment spill)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(Ab), 0, (int)(((int64_t)d * m * 4 > 0x7fffffffLL) ? 0x7fffffff : (int64_t)d * m * 4),
      0x00020000);
  auto frag_offset = [&](int t, bool& iv, int& i) -> int {
    const int tile = blockIdx.x * kMsTiles + t;
    i = tile * 128 + wave * 32 + l31;
    iv = (t < kMsTiles) && (i < m);
    return iv ? (half * m + i) * 4 : 0x7ffffff0;  // out of range -> the loads return 0
  };
  auto load_k = [&](int voff, int kk) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, kk * 2 * m * 4, 0));
  };
  float xf[DH], xn[DH];  // this tile's data fragment (B operand), and the next tile's in flight
  bool iv, ivn = false;
  int i, in_ = 0;
  {
    const int voff = frag_offset(0, iv, i);
#pragma unroll
    for (int kk = 0; kk < DH; ++kk) xf[kk] = load_k(voff, kk);
  }

  // A tile = 8 UNITS of 32 centroids; a unit is one chain of DH MFMAs into a single 32x32
  // accumulator (back-to-back accumulation into the same registers is forwarded by the matrix
  // pipe).  Units alternate between two accumulators: while unit u runs, the 16 values per lane
  // of unit u-1 go through the arg-max epilogue (2 acc - |a|^2 - |b|^2, compare, select) in
  // slices BETWEEN the MFMAs, as do the |a|^2 chain of the tile (under unit 0) and the loads of
  // the next tile's fragment (spread over all 8 units), so the matrix pipe never waits for them.
  // (The previous form -- 4 independent accumulators per pass, epilogue on its own after each
  // pass -- spent 3.5 of 19.9 ms at C5 in the epilogue and 1.3 in |a|^2, measured by knocking
  // them out; a phase-pipelined version of THAT form needed two 64-register accumulator sets and
  // spilled.)  The A operands (centroid rows from LDS) are fetched two k-steps ahead.
  // Within a lane the centroid index grows with (unit, r), so "first maximum" == smallest index.
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = 0.f;
  float best = -INFINITY;
  int besti = 0, bestu = 0;  // index within the unit (0..27, without the lane's 4*half), unit
  float a2_prev = 0.f;
  bool iv_prev = false;
  int i_prev = 0;

  // value r (0..15) of finished unit fu -> running (best, besti).  besti is tracked WITHOUT the
  // lane's 4*half offset: the candidate index is then an instruction literal (with the offset in
  // it the compiler hoists loop-invariant index registers out of the tile loop and spills)
  const float* b2h = b2s + 4 * half;
  // The fp32 MFMA shares the SIMD's fp32 ALUs with the VALU (knock-out measurements: VALU work
  // placed between the MFMAs is NOT hidden, it adds), so the epilogue is counted in instructions:
  // values go in PAIRS (r, r+1: consecutive centroids) through v_pk_fma_f32 (2 acc - |a|^2: the
  // doubling is exact, so this is the two-step 2*acc, then -|a|^2) and v_pk_add_f32 (-|b|^2 pair,
  // one ds_read_b64), then compare + two selects each with the centroid index as an instruction
  // literal: 4 VALU instructions per value instead of 7.
  auto cl_of = [](int fu, int r) { return fu * 32 + (r & 3) + 8 * (r >> 2); };
  auto epi_pair = [&](const f32x16& fin, auto fu_c, auto r_c, float a2, f32x2 b2) {
    constexpr int r = decltype(r_c)::value;
    f32x2 v = {fin[r], fin[r + 1]};
    if (euclidean) {
      const f32x2 two = {2.f, 2.f}, na2 = {-a2, -a2};
      v = __builtin_elementwise_fma(v, two, na2);
      v = v - b2;
    } else {
      v = v + b2;
    }
    ms_take<(r & 3) + 8 * (r >> 2)>(best, besti, v[0]);
    ms_take<((r + 1) & 3) + 8 * ((r + 1) >> 2)>(best, besti, v[1]);
  };
  auto finish_tile = [&](bool fiv, int fi) {
    besti += 32 * bestu + 4 * half;
    const float ov = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(besti, 32, 64);
    if (ov > best || (ov == best && oi < besti)) {
      best = ov;
      besti = oi;
    }
    if (half == 0 && fiv) {
      besti += c0;
      if (!first) {
        const float pv = vals[(int64_t)b * m + fi];
        if (!(best > pv)) {
          best = pv;
          besti = (int)inds[(int64_t)b * m + fi];
        }
      }
      vals[(int64_t)b * m + fi] = best;
      inds[(int64_t)b * m + fi] = besti;
    }
  };
  constexpr int PF = DH >= 2 ? 2 : 1;  // A-operand prefetch distance (k-steps)
  const int row2 = 2 * m * 4;  // bytes between the k-rows a lane owns
  // unit U of the current tile into `acc`; `fin` = the accumulator of the unit before it
  auto unit = [&](auto u_c, f32x16& acc, const f32x16& fin, float a2_fin, float& a2_out, int& voff_next) {
    constexpr int U = decltype(u_c)::value, FU = (U + 7) & 7;
    const float* crow = cs + half * 256 + U * 32 + l31;
    float ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) ring[p] = crow[p * 512];
    float a2 = 0.f;
    // |b|^2 pairs are read one slice ahead: no LDS latency inside a slice
    f32x2 b2n = *reinterpret_cast<const f32x2*>(b2h + cl_of(FU, 0));
    const float best_before = best;  // the epilogue of unit FU starts here
    static_for<0, DH>([&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
      const float a_cur = ring[kk % PF];
      if (kk + PF < DH) ring[kk % PF] = crow[(kk + PF) * 512];
      if (kk == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, xf[kk], zero, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, xf[kk], acc, 0, 0, 0);
      }
      if constexpr (U == 0 && euclidean) {
        // |a|^2 as the ascending-k fma chain: this lane holds k = 2kk+half, its partner
        // (lane ^ 32) the other parity; one half-swap hands every lane both (a k beyond d reads 0).
        // (inline asm, hazard wait states inside the string: with the builtin hipcc 7.2 used
        // result[0] for both results once float math followed -- tools/ubench/permlane_swap.hip)
        float x0 = xf[kk], x1 = xf[kk];
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
        a2 = fmaf(x0, x0, a2);
        a2 = fmaf(x1, x1, a2);
      }
      // next tile's fragment: DH loads spread evenly over the 8 DH k-steps of the tile
      // (running per-lane offset, no soffset: DH distinct row offsets would be hoisted out of
      // the tile loop as DH SGPRs and spilled to VGPR lanes)
      if constexpr ((U * DH + kk) % 8 == 0) {
        xn[(U * DH + kk) / 8] =
            __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_next, 0, 0));
        voff_next += row2;
      }
      // epilogue slice of the previous unit: its 8 value pairs spread over the DH k-steps
      static_for<(kk * 8) / DH, ((kk + 1) * 8) / DH>([&](auto pr_c) {
        constexpr int r = 2 * decltype(pr_c)::value;
        const f32x2 b2c = b2n;
        if constexpr (r + 2 < 16) b2n = *reinterpret_cast<const f32x2*>(b2h + cl_of(FU, r + 2));
        epi_pair(fin, std::integral_constant<int, FU>{}, std::integral_constant<int, r>{}, a2_fin, b2c);
      });
      // keep the scheduler from hoisting every k-step's LDS reads to the top
      __builtin_amdgcn_sched_barrier(0);
    });
    bestu = best > best_before ? FU : bestu;  // unit FU's epilogue is complete
    a2_out = a2;
  };
  using std::integral_constant;

  bool have_prev = false;
#pragma unroll 1
  for (int t = 0; t < kMsTiles; ++t) {
    if ((blockIdx.x * kMsTiles + t) * 128 >= m) break;
    int voff_next = frag_offset(t + 1, ivn, in_);
    // (lanes without a point start at 0x7ffffff0; wherever their DH steps of row2 land, buffer
    // loads are range-checked and the lane's result is never written)
    // unit 0: under it, the epilogue of the LAST unit of tile t-1 (a zero accumulator and a
    // discarded result for the first tile) and |a|^2 of tile t
    float a2_t = 0.f, dummy = 0.f;
    unit(integral_constant<int, 0>{}, accA, accB, a2_prev, a2_t, voff_next);
    if (have_prev) finish_tile(iv_prev, i_prev);
    best = -INFINITY;
    besti = 0;
    bestu = 0;
    unit(integral_constant<int, 1>{}, accB, accA, a2_t, dummy, voff_next);
    unit(integral_constant<int, 2>{}, accA, accB, a2_t, dummy, voff_next);
    unit(integral_constant<int, 3>{}, accB, accA, a2_t, dummy, voff_next);
    unit(integral_constant<int, 4>{}, accA, accB, a2_t, dummy, voff_next);
    unit(integral_constant<int, 5>{}, accB, accA, a2_t, dummy, voff_next);
    unit(integral_constant<int, 6>{}, accA, accB, a2_t, dummy, voff_next);
    unit(integral_constant<int, 7>{}, accB, accA, a2_t, dummy, voff_next);
    a2_prev = a2_t;
    iv_prev = iv;
    i_prev = i;
    have_prev = true;
#pragma unroll
    for (int kk = 0; kk < DH; ++kk) xf[kk] = xn[kk];
    iv = ivn;
    i = in_;
  }
  if (have_prev) {  // epilogue of the last unit of the last tile: nothing left to hide it under
    const float best_before = best;
    static_for<0, 8>([&](auto pr_c) {
      constexpr int r = 2 * decltype(pr_c)::value;
      epi_pair(accB, std::integral_constant<int, 7>{}, std::integral_constant<int, r>{}, a2_prev,
               *reinterpret_cast<const f32x2*>(b2h + cl_of(7, r)));
    });
    bestu = best > best_before ? 7 : bestu;
    finish_tile(iv_prev, i_prev);
  }
}

template <int DH>
static int launch_codebook(const float* A, const float* B, float* vals, int64_t* inds, int l, int d,
                           int m, int n, int euclid, hipStream_t st) {
  const size_t lds = (size_t)(2 * DH * 256 + 256) * sizeof(float);
  const dim3 grid((m + 128 * kMsTiles - 1) / (128 * kMsTiles), l);
  auto go = [&](auto kernel) -> int {
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "max_sim_codebook_kernel attr");
    if (rc) return rc;
    for (int c0 = 0; c0 < n; c0 += 256) {
      hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, A, B, vals, inds, d, m, n, c0,
                         c0 == 0 ? 1 : 0);
      TPQ_LAUNCH_CHECK("max_sim_codebook_kernel");
    }
    return TPQ_OK;
  };
  return euclid ? go(max_sim_codebook_kernel<DH, true>) : go(max_sim_codebook_kernel<DH, false>);
}

// ---- update --------------------------------------------------------------------------------
constexpr int kCcDT = 16;        // dimensions per block

// grid (ceil(n/points), ceil(d/kCcDT), l); LDS: [kCcDT][k] sums + [k] counts
__global__ __launch_bounds__(256) void centroid_accum_kernel(const float* __restrict__ data,
                                                             const int64_t* __restrict__ labels,
                                                             float* __restrict__ sums,
                                                             float* __restrict__ counts, int d,
                                                             int64_t n, int k, int64_t points) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* ssum = sh;               // [kCcDT][k]
  float* scnt = sh + kCcDT * k;   // [k]
  const int b = blockIdx.z;
  const int e0 = blockIdx.y * kCcDT;
  const int ne = (d - e0) < kCcDT ? (d - e0) : kCcDT;
  for (int t = threadIdx.x; t < (kCcDT + 1) * k; t += 256) sh[t] = 0.f;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * points;
  const int64_t i1 = (i0 + points) < n ? (i0 + points) : n;
  const bool count_here = (blockIdx.y == 0);
  const float* __restrict__ drow = data + ((int64_t)b * d + e0) * n;
  const int64_t* __restrict__ lrow = labels + (int64_t)b * n;
  if (ne == kCcDT && (n & 3) == 0) {
    // full 16-dimension tile, 4 points per thread: 16 independent 16-byte loads in flight per
    // thread before the first LDS atomic (the scalar loop below is latency-bound: one 4-byte
    // load per ds_add)
    for (int64_t i = i0 + (int64_t)threadIdx.x * 4; i < i1; i += 256 * 4) {
      float4 x[kCcDT];
#pragma unroll
      for (int e = 0; e < kCcDT; ++e) x[e] = *reinterpret_cast<const float4*>(drow + (int64_t)e * n + i);
      const longlong2 la = *reinterpret_cast<const longlong2*>(lrow + i);
      const longlong2 lb = *reinterpret_cast<const longlong2*>(lrow + i + 2);
      const long long lab[4] = {la.x, la.y, lb.x, lb.y};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i + u >= i1 || lab[u] < 0 || lab[u] >= k) continue;
        if (count_here) atomicAdd(&scnt[lab[u]], 1.0f);
#pragma unroll
        for (int e = 0; e < kCcDT; ++e) {
          const float v = u == 0 ? x[e].x : u == 1 ? x[e].y : u == 2 ? x[e].z : x[e].w;
          atomicAdd(&ssum[e * k + lab[u]], v);
        }
      }
    }
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
      const int64_t lab = lrow[i];
      if (lab < 0 || lab >= k) continue;
      if (count_here) atomicAdd(&scnt[lab], 1.0f);
      for (int e = 0; e < ne; ++e) atomicAdd(&ssum[e * k + lab], drow[(int64_t)e * n + i]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < ne * k; t += 256) {
    const float s = ssum[t];
    if (s != 0.f) unsafeAtomicAdd(&sums[((int64_t)b * d + e0) * k + t], s);
  }
  if (count_here)
    for (int t = threadIdx.x; t < k; t += 256) {
      const float c = scnt[t];
      if (c != 0.f) unsafeAtomicAdd(&counts[(int64_t)b * k + t], c);
    }
}

// ---- update, many clusters (coarse quantiser: k in the thousands) ------------------------------
// With thousands of bins per dimension an LDS privatisation no longer fits and contention on any
// one bin is low, so each (point, dimension) goes straight to an L2 float atomic.
__global__ __launch_bounds__(256) void centroid_accum_global_kernel(
    const float* __restrict__ data, const int64_t* __restrict__ labels, float* __restrict__ sums,
    float* __restrict__ counts, int d, int64_t n, int k) {
  const int b = blockIdx.z;
  const int e0 = blockIdx.y * kCcDT;
  const int ne = (d - e0) < kCcDT ? (d - e0) : kCcDT;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t lab = labels[(int64_t)b * n + i];
  if (lab < 0 || lab >= k) return;
  if (blockIdx.y == 0) unsafeAtomicAdd(&counts[(int64_t)b * k + lab], 1.0f);
  for (int e = 0; e < ne; ++e)
    unsafeAtomicAdd(&sums[((int64_t)b * d + e0 + e) * k + lab], data[((int64_t)b * d + e0 + e) * n + i]);
}

// ---- update on the bf16 matrix cores (k <= 256) ---------------------------------------------
// sums[cluster][dim] = sum_i onehot(label_i)[cluster] * x_i[dim] is a GEMM whose left operand is a
// 0/1 matrix.  fp32 MFMA would waste 255/256 of its multiplies at 1/16 of the bf16 rate; instead x
// is split EXACTLY into three bf16 pieces (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi -
// mid): 3 x 8 significant bits), the one-hot tile is built in registers from the labels, and
// v_mfma_f32_32x32x16_bf16 accumulates 1.0 * piece products in fp32 -- exact products, fp32 sums,
// 16 points per instruction.  Operand layout (tools/ubench/mfma_bf16_layout.hip): A[row][k] in
// lane row + 32 (k / 8), element k % 8; B[k][col] likewise; D as the f32 forms.
// LDS float atomics are the trap on the scalar route: ds_add_f32 retires ~0.38 lanes per clock
// per CU on gfx950 whatever the access pattern (tools/ubench/lds_atomic.hip; integer atomics are
// 16x faster); an atomic-free LDS read-add-write version reached 9.2 ms at C5, this one 4.8 ms.
// (A NaN / Inf coordinate reaches every cluster of its 16-point group through 0 * x; the scalar
// kernels confine it to its own cluster.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kUmP = 64;              // points per staged tile (4 MFMA k-steps of 16)
constexpr int kUmStride = kUmP + 4;   // floats per dimension row in LDS (b128 reads stay conflict-free)

// One WAVE per block owns all 256 clusters x 32*CT dimensions of its tiles: 8 x CT accumulator
// tiles.  CT = 2 (64 dimensions, 256 accumulator registers) leaves one wave per SIMD: its VALU
// work and its tile staging then run in series with its own MFMAs (5.9 ms at C5 against 2.5 ms of
// matrix-pipe time).  CT = 1 (32 dimensions, 128 registers) puts two independent waves on every
// SIMD -- no barrier between them, each splits only its own dimensions -- so one wave's VALU / LDS
// / load phases sit under the other's MFMAs.  (A first version spread the CLUSTERS over the 4
// waves of a block: every wave then split the same tile into bf16 pieces again and the block met
// at a barrier per 64 points -- 7.4 ms at C5; the LDS read-add-write kernel above: 9.2 ms.)
// The B fragment wants 8 consecutive points of ONE dimension per lane; straight from global memory
// that is one 32-byte request per lane (address-unit bound, 32 ms), so [dims][64 points] tiles
// go through LDS, loaded two tiles ahead (below).
// r02 at C5: 5.9 -> 4.8 ms (3.5 TB/s; the bare read pattern streams at 6.2 TB/s --
// tpq_ubench_rows_read -- and the MFMAs need 2.5 ms; what is left is a wave waiting, 59 % of its
// cycles by SQ_WAIT_INST_ANY, for its own LDS round trips at the head of every k-step: two waves
// per SIMD do not cover them, and a second one-hot table to pipeline k-steps does not fit 20 KiB
// of LDS per wave).
#ifndef TPQ_UM_CT
#define TPQ_UM_CT 1
#endif
#ifndef TPQ_UM_EXP
#define TPQ_UM_EXP 0  // experiments (tools/build_variant.sh): 1 = no MFMAs, 2 = no tile reloads
#endif
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// (Lesson kept from the version that built the one-hot operand in registers with three
// packed-u16 instructions per label pair: written as inline asm they produced scheduling-dependent
// WRONG sums -- the hazard recogniser cannot see a VALU write inside an asm block that an MFMA reads
// as its A operand a few cycles later; the same instructions selected by the compiler from plain
// vector code (__builtin_elementwise_sub_sat, u16x2 multiply) were correct.)

template <int CT, bool VEC>
__global__ __launch_bounds__(64, (CT == 1 ? 2 : 1)) void centroid_accum_mfma_kernel(
    const float* __restrict__ data, const int64_t* __restrict__ labels, float* __restrict__ sums,
    float* __restrict__ counts, int d, int64_t n, int k) {
  constexpr int ND = 32 * CT;  // dimensions per wave
  // one wave per block: its DS operations execute in order, so ONE tile buffer is enough (the
  // stores of tile t+1 queue up behind the reads of tile t) and no barrier is ever needed
  __shared__ __attribute__((aligned(16))) float xt[ND * kUmStride];
  // the one-hot operand of the current k-step: [2 k-groups][256 clusters][8 points] bf16, all zero
  // except one 1.0 per point (see below)
  __shared__ __attribute__((aligned(16))) uint16_t otab[256 * 16];
  __shared__ int cnt[256];
  const int b = blockIdx.z;
  const int lane = threadIdx.x;
  const int l31 = lane & 31, half = lane >> 5;
  const int e0 = blockIdx.y * ND;
  const int nd = (d - e0) < ND ? (d - e0) : ND;  // dimensions of this block that exist
  // Tiles are dealt round-robin to the gridDim.x blocks of a (sub-problem, dimension tile): the
  // blocks run side by side, so at any moment they read ADJACENT 256-byte pieces of the same
  // 32 rows -- whole DRAM pages get used while they are open.  (With one contiguous point range
  // per block every 256-byte access opened a page of its own: the kernel read at 3.4 TB/s.)
  const int64_t step = (int64_t)gridDim.x * kUmP;
  const int64_t i0 = (int64_t)blockIdx.x * kUmP;  // first tile of this block (exists: host)
  const int64_t i1 = n;
  const int64_t* __restrict__ lrow = labels + (int64_t)b * n;
  const float* __restrict__ dbase = data + ((int64_t)b * d + e0) * n;
  f32x16 acc[8][CT];  // [cluster row tile][dimension column tile]
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) cnt[lane + 64 * u] = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) reinterpret_cast<u32x4v*>(otab)[lane + 64 * u] = u32x4v{0u, 0u, 0u, 0u};
  const bool count_here = blockIdx.y == 0;

  // Tiles travel global -> registers -> LDS, TWO tiles ahead of the MFMAs: the loads of tile t+2
  // are issued in four quarters, one per MFMA k-step of tile t (so the wave's load issue hides
  // behind its MFMAs), and reach LDS at the end of tile t+1 -- more than two tile times in flight
  // (with one tile ahead the wave waited out part of every memory latency: 5.65 ms at C5 against
  // 3.7 ms with the reloads knocked out).  VEC (n % 4 == 0): a lane loads 4 consecutive points of
  // one dimension row (16 B), a wave-instruction 4 rows x 256 B; otherwise one point per lane.
  // Loads are unconditional on clamped addresses: out-of-range points carry label 0xFFFF (their
  // one-hot column is zero) and out-of-range dimension rows land in accumulator columns that are
  // never written out; no exec-mask branches, no selects behind the loads.
  constexpr int NV = VEC ? ND / 4 : ND;  // load instructions (registers: NV float4 / NV floats)
  typedef typename std::conditional<VEC, f32x4, float>::type xreg_t;
  struct Staged {
    xreg_t x[NV];
    int64_t label;
    bool label_valid;
  };
  const int vrow = VEC ? (lane >> 4) : 0, vpt = VEC ? (lane & 15) * 4 : lane;
  auto load_quarter = [&](Staged& st, int64_t p0, int qt) {
    const int64_t pt = p0 + vpt;
    const bool pv = pt < i1;
    const float* __restrict__ p = dbase + (pv ? pt : i0);
#pragma unroll
    for (int j = (NV / 4) * qt; j < (NV / 4) * (qt + 1); ++j) {
      const int row = VEC ? 4 * j + vrow : j;
      st.x[j] = *reinterpret_cast<const xreg_t*>(p + (int64_t)(row < nd ? row : 0) * n);
    }
    if (qt == 0) {  // the raw label: any arithmetic on it here would wait for the load on the spot
      const int64_t lp = p0 + lane;
      st.label_valid = lp < i1;
      st.label = lrow[st.label_valid ? lp : i0];
    }
  };
  int lab_cur = -1;  // label of point (tile in LDS) + lane, -1 = none
  auto store_tile = [&](const Staged& st) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int row = VEC ? 4 * j + vrow : j;
      *reinterpret_cast<xreg_t*>(&xt[row * kUmStride + vpt]) = st.x[j];
    }
    lab_cur = (st.label_valid && st.label >= 0 && st.label < k) ? (int)st.label : -1;
  };
  // one tile: MFMAs from the LDS tile; `fill` receives tile it+2; `drain` (tile it+1) replaces the
  // LDS tile afterwards.
  // The one-hot A operand costs NO VALU work: the [256 clusters][16 points] bf16 matrix of a
  // k-step lives in LDS, all zero; the 16 lanes that own the k-step's points each store one 1.0
  // at [label][point] (ds_write_b16), every lane then reads its 8 row tiles as ds_read_b128 --
  // 16 contiguous bytes = the 8 consecutive points of its k-group, exactly the fragment -- and the
  // writers store the zero back.  (Built in registers from packed label pairs the operand took
  // 96 VALU instructions per k-step, 3 per pair and row tile, next to 44 for the bf16 splitting:
  // the two waves of a SIMD then issue as many VALU cycles as MFMA cycles and the update ran at
  // 5.2 ms against 2.5 ms of matrix-pipe time.)
  auto run_tile = [&](int64_t p0, Staged& fill, const Staged& drain) {
    // (no "is there a tile t+1 / t+2" branches: beyond the range the loads read clamped addresses
    // and the store fills a tile nobody reads.  With conditional loads the compiler's waitcnt
    // bookkeeping merges the two paths and falls back to vmcnt(0) before the LDS stores, i.e. it
    // waits for the loads of tile t+2 that were only just issued)
    if (count_here && lab_cur >= 0) atomicAdd(&cnt[lab_cur], 1);  // integer LDS atomic: fast
    // layout [k-group (2)][cluster (256)][8 points]: the 16 lanes of a ds_read_b128 group then read
    // 256 contiguous bytes (with [cluster][16 points] rows two lanes of a group met on a bank:
    // SQ_LDS_BANK_CONFLICT was 36 % of the LDS cycles)
    uint16_t* oslot = &otab[((lane >> 3) & 1) * 2048 + (lab_cur >= 0 ? lab_cur : 0) * 8 + (lane & 7)];
#pragma unroll
    for (int ks = 0; ks < kUmP / 16; ++ks) {
      if (!(TPQ_UM_EXP & 2)) load_quarter(fill, p0 + 2 * step, ks);
      const bool writer = (lane >> 4) == ks && lab_cur >= 0;
      if (writer) *oslot = (uint16_t)0x3F80;  // bf16 1.0
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int pts = 16 * ks + 8 * half;  // this lane's 8 points (its k-group)
      const bf16x8* orow = reinterpret_cast<const bf16x8*>(&otab[half * 2048 + l31 * 8]);
      bf16x8 aring[3];  // A operands are fetched two row tiles ahead of their MFMAs
      aring[0] = orow[0];
      aring[1] = orow[32];
      bf16x8 piece[3][CT];  // [hi, mid, lo][column tile]
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float* xrow = &xt[(32 * ct + l31) * kUmStride + pts];
        const float4 xa = *reinterpret_cast<const float4*>(xrow);
        const float4 xb = *reinterpret_cast<const float4*>(xrow + 4);
        const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const __bf16 h = (__bf16)xv[i];
          const float r1 = xv[i] - (float)h;
          const __bf16 m = (__bf16)r1;
          const float r2 = r1 - (float)m;
          piece[0][ct][i] = h;
          piece[1][ct][i] = m;
          piece[2][ct][i] = (__bf16)r2;
        }
      }
      // the accumulator tiles take turns: an MFMA never waits for the one issued before it
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
        if (rt + 2 < 8) aring[(rt + 2) % 3] = orow[32 * (rt + 2)];  // 32 rows x 16 B
        const bf16x8 aop = aring[rt % 3];
        // (all column tiles always: a wave-uniform branch around the second one when d <= 32
        // broke the MFMA interleaving and cost more than the multiplies it saved)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            if (TPQ_UM_EXP & 1) {
              acc[rt][ct][pc] += (float)aop[pc] + (float)piece[pc][ct][rt];
              continue;
            }
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aop, piece[pc][ct], acc[rt][ct], 0, 0, 0);
          }
      }
      __builtin_amdgcn_wave_barrier();
      if (writer) *oslot = (uint16_t)0;  // after the last read of this k-step has been issued
    }
    store_tile(drain);
    // the next tile's reads see these stores without a barrier (in-order DS) -- and a
    // __syncthreads() would bring an s_waitcnt vmcnt(0) with it, i.e. wait for the loads of tile
    // t+2 that were only just issued
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  Staged sa, sb;
  sa.label = sb.label = -1;
  sa.label_valid = sb.label_valid = false;
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) load_quarter(sa, i0, qt);
  store_tile(sa);
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) load_quarter(sa, i0 + step, qt);  // clamped when out of range
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll 1
  for (int64_t p0 = i0; p0 < i1; p0 += 2 * step) {
    run_tile(p0, sb, sa);                             // tile 2j: fill sb (2j+2), drain sa (2j+1)
    if (p0 + step < i1) run_tile(p0 + step, sa, sb);  // tile 2j+1: fill sa (2j+3), drain sb (2j+2)
  }
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int dim = e0 + 32 * ct + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cluster = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = acc[rt][ct][r];
        if (dim < d && cluster < k && v != 0.f)
          unsafeAtomicAdd(&sums[((int64_t)b * d + dim) * k + cluster], v);
      }
    }
  if (count_here) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = lane + 64 * u;
      if (c < k && cnt[c]) unsafeAtomicAdd(&counts[(int64_t)b * k + c], (float)cnt[c]);
    }
  }
}

__global__ __launch_bounds__(256) void centroid_finalize_kernel(const float* __restrict__ sums,
                                                               const float* __restrict__ counts,
                                                               float* __restrict__ out, int d, int k,
                                                               int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % k);
  const int64_t b = t / ((int64_t)d * k);
  const float cnt = counts[b * k + c];
  out[t] = cnt == 0.f ? 0.f : sums[t] / cnt;  // compute_centroids.cu:82
}

}  // namespace tpq

using namespace tpq;

extern "C" int tpq_max_sim(const float* A, const float* B, float* vals, int64_t* inds, int l, int d,
                           int m, int n, int metric, tpq_stream_t stream) {
  TPQ_REQUIRE(A && B && vals && inds, "max_sim: null pointer");
  TPQ_REQUIRE(l >= 1 && d >= 1 && m >= 0 && n >= 1, "max_sim: bad shape l=%d d=%d m=%d n=%d", l, d, m, n);
  TPQ_REQUIRE(metric == TPQ_METRIC_NEG_SQ_L2 || metric == TPQ_METRIC_INNER, "max_sim: bad metric %d", metric);
  TPQ_REQUIRE(l <= 65535, "max_sim: batch l=%d exceeds grid.y", l);
  if (m == 0) return TPQ_OK;
  const int euclid = metric == TPQ_METRIC_NEG_SQ_L2 ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // centroids resident in LDS, 256 at a time (one launch per chunk; PQ codebooks need one)
  // (beyond one chunk of centroids the double-buffered generic kernel wins from d > 64 on:
  // 1 M x 1024: d=64 84 vs 62 TF/s, d=96 47 vs 71, d=128 60 vs 78)
  // (the codebook kernel addresses a sub-problem's data through one buffer resource with 32-bit
  // offsets: slices of 2 GiB or more -- d * m * 4 bytes -- go to the generic kernel's 64-bit pointers)
  if (d <= 128 && n <= 65536 && (n <= 256 || d <= 64) && (int64_t)d * m * 4 <= 0x7fffffffLL) {
    const int dh = (d + 1) / 2;
    if (dh <= 1) return launch_codebook<1>(A, B, vals, inds, l, d, m, n, euclid, st);
    if (dh <= 2) return launch_codebook<2>(A, B, vals, inds, l, d, m, n, euclid, st);
    if (dh <= 4) return launch_codebook<4>(A, B, vals, inds, l, d, m, n, euclid, st);
    if (dh <= 8) return launch_codebook<8>(A, B, vals, inds, l, d, m, n, euclid, st);
    if (dh <= 16) return launch_codebook<16>(A, B, vals, inds, l, d, m, n, euclid, st);
    if (dh <= 32) return launch_codebook<32>(A, B, vals, inds, l, d, m, n, euclid, st);
    return launch_codebook<64>(A, B, vals, inds, l, d, m, n, euclid, st);
  }
  const size_t ms_lds = (size_t)(2 * kMsSlab + kMsCent) * sizeof(float);
  int rc_attr = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(max_sim_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ms_lds),
                          "max_sim_kernel attr");
  if (rc_attr) return rc_attr;
  hipLaunchKernelGGL(max_sim_kernel, dim3((m + 127) / 128, l), dim3(256), ms_lds, st, A, B, vals,
                     inds, d, m, n, euclid, static_cast<const int*>(nullptr),
                     static_cast<const int*>(nullptr), static_cast<unsigned long long*>(nullptr),
                     static_cast<const float*>(nullptr), 0, 0, 0);
  TPQ_LAUNCH_CHECK("max_sim_kernel");
  return TPQ_OK;
}

extern "C" size_t tpq_compute_centroids_workspace_bytes(int l, int d, int k) {
  return ((size_t)l * d * k + (size_t)l * k) * sizeof(float);
}

extern "C" int tpq_compute_centroids(const float* data, const int64_t* labels, float* centroids,
                                     int l, int d, int64_t n, int k, void* workspace,
                                     size_t workspace_bytes, tpq_stream_t stream) {
  TPQ_REQUIRE(data && labels && centroids, "compute_centroids: null pointer");
  TPQ_REQUIRE(l >= 1 && d >= 1 && n >= 0 && k >= 1, "compute_centroids: bad shape");
  TPQ_REQUIRE(l <= 65535, "compute_centroids: batch l=%d exceeds grid.z", l);
  const size_t need = tpq_compute_centroids_workspace_bytes(l, d, k);
  if (!workspace || workspace_bytes < need) {
    set_error("compute_centroids: workspace too small (%zu < %zu)", workspace_bytes, need);
    return TPQ_ERR_WORKSPACE;
  }
  const size_t lds = (size_t)(kCcDT + 1) * k * sizeof(float);
  const bool lds_fits = lds <= 160 * 1024;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = check_hip(hipMemsetAsync(workspace, 0, need, st), "compute_centroids memset");
  if (rc) return rc;
  float* sums = reinterpret_cast<float*>(workspace);
  float* counts = sums + (size_t)l * d * k;
  if (n > 0) {
    if (lds_fits)
      rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(centroid_accum_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                   "centroid_accum_kernel attr");
    if (rc) return rc;
    if (k <= 256 && d >= 32) {  // wide PQ-codebook shape: bf16 matrix cores
      constexpr int CT = TPQ_UM_CT;
      const int dtiles = (d + 32 * CT - 1) / (32 * CT);
      // blocks per (sub-problem, dimension tile): a few rounds of the 2048 / CT resident waves,
      // but at least 64 tiles (4096 points) per block -- every block ends with 256 x 32 CT global
      // atomics -- and never more blocks than tiles
      int64_t chunks = (4096 / CT) / ((int64_t)l * dtiles);
      const int64_t n_tiles = (n + kUmP - 1) / kUmP;
      if (chunks > n_tiles / 64) chunks = n_tiles / 64;
      if (chunks < 1) chunks = 1;
      const dim3 grid((unsigned)chunks, dtiles, l);
      if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(data) & 15) == 0)
        hipLaunchKernelGGL((centroid_accum_mfma_kernel<CT, true>), grid, dim3(64), 0, st, data, labels,
                           sums, counts, d, n, k);
      else
        hipLaunchKernelGGL((centroid_accum_mfma_kernel<CT, false>), grid, dim3(64), 0, st, data, labels,
                           sums, counts, d, n, k);
      TPQ_LAUNCH_CHECK("centroid_accum_mfma_kernel");
      const int64_t total = (int64_t)l * d * k;
      hipLaunchKernelGGL(centroid_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                         0, st, sums, counts, centroids, d, k, total);
      TPQ_LAUNCH_CHECK("centroid_finalize_kernel");
      return TPQ_OK;
    }
    if (!lds_fits) {
      hipLaunchKernelGGL(centroid_accum_global_kernel,
                         dim3((unsigned)((n + 255) / 256), (d + kCcDT - 1) / kCcDT, l), dim3(256), 0,
                         st, data, labels, sums, counts, d, n, k);
      TPQ_LAUNCH_CHECK("centroid_accum_global_kernel");
      const int64_t total = (int64_t)l * d * k;
      hipLaunchKernelGGL(centroid_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                         0, st, sums, counts, centroids, d, k, total);
      TPQ_LAUNCH_CHECK("centroid_finalize_kernel");
      return TPQ_OK;
    }
    // Each block flushes kCcDT*k global atomics, so blocks must own many points; aim for ~8 k
    // blocks in total (>> 256 CUs) but never fewer than 4096 points per block.
    const int dtiles = (d + kCcDT - 1) / kCcDT;
    int64_t chunks = 8192 / ((int64_t)l * dtiles);
    if (chunks < 1) chunks = 1;
    int64_t points = (n + chunks - 1) / chunks;
    if (points < 4096) points = 4096;
    points = (points + 255) / 256 * 256;
    hipLaunchKernelGGL(centroid_accum_kernel, dim3((unsigned)((n + points - 1) / points), dtiles, l),
                       dim3(256), lds, st, data, labels, sums, counts, d, n, k, points);
    TPQ_LAUNCH_CHECK("centroid_accum_kernel");
  }
  const int64_t total = (int64_t)l * d * k;
  hipLaunchKernelGGL(centroid_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st, sums, counts, centroids, d, k, total);
  TPQ_LAUNCH_CHECK("centroid_finalize_kernel");
  return TPQ_OK;
}
