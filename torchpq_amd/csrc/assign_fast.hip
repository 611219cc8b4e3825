// Bit-exact assign (labels) at bf16-matrix-core speed: error-bounded selection on a split-bf16 MFMA
// kernel + exact re-check of the ambiguous points.  Two entry points: tpq_coarse_assign (one problem,
// many centroids, points resident: the coarse assign of add) and tpq_max_sim_select (batched
// codebook-sized problems, centroids resident: the PQ codebook training; section 2b below).
//
// tpq_coarse_assign returns, for every point, the SAME label tpq_max_sim returns (the arg-max of the
// oracle's fp32 arithmetic: ascending-k fmaf chains, (2 acc - |a|^2) - |c|^2, ties -> smallest
// index) -- it replaces the max_sim call behind VQCodec.encode / IVFPQIndex.add
// (torchpq/index/IVFPQIndex.py:233-256 -> codec/VQCodec.py -> clustering/KMeans.py:440-452 ->
// kernels/MaxSimCuda.py:296-340, kernel max_sim_tn torchpq/kernels/cuda/max_sim.cu:182-309), which
// is 90 % of an add() at 16 384 cells.  The same idea as the list scan: a FAST value f
// with a rigorous bound |f - g| <= delta on its distance from the real-number value g selects, the
// exact arithmetic decides only where the selection cannot:
//   1. assign_mean_kernel + assign_prep_kernel: the centroids, centred on their mean, doubled (exact), split into NP bf16 pieces (NP = 2:
//      c = c1 + c2 + r, |r| <= 2^-16 |c|: a bf16 piece carries 8 significant bits, unit roundoff
//      2^-8) and laid out in MFMA-fragment order, 32 centroids per unit,
//      plus -|c|^2 as an exact 3-piece fragment and max |c|^2;
//   2. assign_fast_kernel: every wave keeps 32 CT points (split the same way) in registers for the
//      whole sweep and streams ALL centroid units through a double-buffered LDS ring filled by
//      global_load_lds (LDS-DMA: no staging registers); per 16 dimensions and 32 x 32 tile the
//      products c2 a1, c1 a2, c1 a1 (NP = 2) go through v_mfma_f32_32x32x16_bf16; the epilogue keeps
//      the best AND the second-best fast value per point (4 VALU per value: compare, index select,
//      median-of-three, max);
//      delta = 1.25 [(eps_prod + (16 KS + 13) 2^-23) (|a - mu| + |c - mu|max)^2
//                    + (d + 4) 2^-24 (|a| + |c|max)^2]
//      (mu = mean centroid: distances are translation-invariant) covers the dropped products (c2 a2,
//      r_c a, c r_a: 3 x 2^-16 |a_k c_k| per term), a worst-case (truncating, any order) fp32
//      accumulation of all MFMA terms (small products first: see the kernel), the rounding of the
//      shift by mu and -- on the raw norms -- the rounding of the exact chain itself.  A point whose two
//      best fast values are further apart than 2 delta has its label decided: any other centroid is
//      worse in the exact arithmetic too.  The rest (1-3 % at d = 128) are appended to a list;
//   3. the bit-exact fp32-MFMA kernel (max_sim_kernel, kmeans.hip) over the listed points only: it
//      reads the list and its length from device memory (no host round trip; the grid covers the
//      worst case and surplus blocks leave at once).
#include <type_traits>

#include "common.h"

namespace tpq {
// the three-level fp16 cascade of lloyd.hip for one problem with many centroids (euclidean, d <= 128)
int lloyd_assign_supported(int d, int64_t m, int n, int route);
size_t lloyd_assign_workspace_bytes(int d, int64_t m, int n);
size_t lloyd_assign_count_offset(int d, int64_t m, int n);
int lloyd_wide_supported(int d, int64_t m, int n);
int lloyd_assign(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n, int euclid, char* ws,
                 hipStream_t st);
int launch_max_sim_list(const float* A, const float* B, float* vals, int64_t* inds, int l, int d, int m, int n,
                        int euclid, const int* list, const int* count, unsigned long long* keys, float* Ac, int cap,
                        hipStream_t st);  // kmeans.hip
namespace afast {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

__device__ __forceinline__ void split3(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

// top-2 of fast values: (b1, b2, bi) <- v with in-unit index CL (inline constant).  A tie with b1
// keeps the earlier index and makes b2 == b1: the point is then ambiguous by construction.
template <int CL>
__device__ __forceinline__ void take_top2(float& b1, float& b2, int& bi, float v) {
  static_assert(CL >= 0 && CL <= 64, "inline constant");
  // b2' = median(b1, b2, v) (b1 >= b2: v >= b1 -> b1, b2 <= v < b1 -> v, v < b2 -> b2); b1' = max
  asm volatile(
      "v_cmp_ngt_f32 vcc, %3, %0\n\t"
      "v_cndmask_b32 %2, %4, %2, vcc\n\t"
      "v_med3_f32 %1, %0, %1, %3\n\t"
      "v_max_f32 %0, %3, %0"
      : "+v"(b1), "+v"(b2), "+v"(bi)
      : "v"(v), "n"(CL)
      : "vcc");
}

// two values, two INDEPENDENT chains, one block (separate asm statements get a wait state between them)
template <int CL0, int CL1>
__device__ __forceinline__ void take_top2_pair(float& p1, float& p2, int& pi, float& q1, float& q2, int& qi,
                                               float v0, float v1) {
  static_assert(CL0 >= 0 && CL0 <= 64 && CL1 >= 0 && CL1 <= 64, "inline constants");
  asm volatile(
      "v_cmp_ngt_f32 vcc, %6, %0\n\t"
      "v_cndmask_b32 %2, %8, %2, vcc\n\t"
      "v_cmp_ngt_f32 vcc, %7, %3\n\t"
      "v_cndmask_b32 %5, %9, %5, vcc\n\t"
      "v_med3_f32 %1, %0, %1, %6\n\t"
      "v_med3_f32 %4, %3, %4, %7\n\t"
      "v_max_f32 %0, %6, %0\n\t"
      "v_max_f32 %3, %7, %3"
      : "+v"(p1), "+v"(p2), "+v"(pi), "+v"(q1), "+v"(q2), "+v"(qi)
      : "v"(v0), "v"(v1), "n"(CL0), "n"(CL1)
      : "vcc");
}

constexpr int kUnitsPerChunk = 4;  // 128 centroids per LDS buffer
constexpr int kWaves = 8;
constexpr int frags_per_unit(int KS, int NP) { return KS * NP + 1; }  // + the -|c|^2 fragment
constexpr size_t chunk_bytes(int KS, int NP) { return (size_t)kUnitsPerChunk * frags_per_unit(KS, NP) * 1024; }
constexpr int n_products(int NP) { return NP == 2 ? 3 : 6; }

// ---- 1. centroid fragments -------------------------------------------------------------------
// grid = units (32 centroids each), block = 64 lanes: lane (row = lane % 32, k-group = lane / 32)
// holds 8 consecutive dimensions of its centroid per k-step, as the MFMA A operand wants them.
// frags: [unit][1 + KS * NP][64 lanes] x 16 B, fragment 0 = -|c|^2 (euclidean) or 0 (inner) as three
// exact pieces at k = 0, 1, 2; rows beyond n carry -3e38 there: they can never be first or second.
// Centring (euclidean only).  Distances are translation-invariant, so the FAST values are computed on
// points and centroids shifted by mu = the mean centroid: the bound of the fast path scales with
// (|a - mu| + |c - mu|max)^2 instead of (|a| + |c|max)^2 -- 2-4x smaller on non-negative data such as
// SIFT -- while the error of the exact chain, which works on the raw data, keeps the raw norms.
// mu[k] = mean over the centroids of dimension k (one block per dimension); zero for inner product.
// blockIdx.y = sub-problem: B + y d n, mu + y 256
__global__ __launch_bounds__(256) void assign_mean_kernel(const float* __restrict__ B, float* __restrict__ mu,
                                                         int n, int euclid) {
  __shared__ float red[256];
  const int k = blockIdx.x;
  B += (int64_t)blockIdx.y * gridDim.x * n;
  mu += blockIdx.y * 256;
  float s = 0.f;
  if (euclid)
    for (int c = threadIdx.x; c < n; c += 256) s += B[(int64_t)k * n + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) mu[k] = red[0] / (float)n;
}

// cmax2_bits[0] = max |c - mu|^2, cmax2_bits[1] = max |c|^2
template <int KS, int NP>
__global__ __launch_bounds__(64) void assign_prep_kernel(const float* __restrict__ B, bf16x8* __restrict__ frags,
                                                        unsigned* __restrict__ cmax2_bits,
                                                        const float* __restrict__ mu, int d, int n,
                                                        int euclid) {
  const int unit = blockIdx.x, lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  const int c = unit * 32 + l31;
  // blockIdx.y = sub-problem (tpq_max_sim_select): its own centroids, mean, maxima and fragment block
  B += (int64_t)blockIdx.y * d * n;
  mu += blockIdx.y * 256;
  cmax2_bits += blockIdx.y * 2;
  frags += (size_t)blockIdx.y * gridDim.x * frags_per_unit(KS, NP) * 64;
  bf16x8* out = frags + (size_t)unit * frags_per_unit(KS, NP) * 64 + lane;
  float s = 0.f, sraw = 0.f;  // |c - mu|^2 of the shifted centroid the fast path uses, and |c|^2
  if (c < n)
    for (int k = 0; k < d; ++k) {
      const float xr = B[(int64_t)k * n + c];
      const float x = xr - mu[k];
      s = fmaf(x, x, s);
      sraw = fmaf(xr, xr, sraw);
    }
  {
    bf16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) {
      __bf16 h, mm, lo;
      split3(c < n ? (euclid ? -s : 0.f) : -3.0e38f, h, mm, lo);
      f[0] = h;
      f[1] = mm;
      f[2] = lo;
    }
    out[0] = f;
  }
  if (half == 0 && c < n) {  // non-negative floats: bit order == value order
    atomicMax(cmax2_bits, __float_as_uint(s));
    atomicMax(cmax2_bits + 1, __float_as_uint(sraw));
  }
#pragma unroll
  for (int st = 0; st < KS; ++st) {
    bf16x8 p[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * st + 8 * half + j;
      float x = (k < d && c < n) ? B[(int64_t)k * n + c] - mu[k] : 0.f;
      if (euclid) x *= 2.f;
      __bf16 h, mm, lo;
      split3(x, h, mm, lo);
      p[0][j] = h;
      p[1][j] = mm;
      p[2][j] = lo;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) out[(1 + st * NP + q) * 64] = p[q];
  }
}

// ---- 2. fast top-2 ---------------------------------------------------------------------------
struct FastArgs {
  const float* A;        // [d][m]
  const bf16x8* frags;   // assign_prep_kernel
  const unsigned* cmax2_bits;  // [0] max |c - mu|^2, [1] max |c|^2
  const float* mu;             // [16 KS] centring vector (zero beyond d and for inner product)
  int64_t* inds;         // [m] fast label (final for unambiguous points)
  float* vals;           // optional [m]: the fast maximum, b1 - |a|^2 (|error| <= delta); exact for re-checked points
  int* list;             // [m] ambiguous points
  int* count;            // their number
  int d, m, n_units, euclid;
  float eps;             // fast path: eps_prod + (terms + 8) 2^-23, relative to (|a - mu| + |c - mu|max)^2
  float eps_exact;       // the exact chain's own rounding: (d + 4) 2^-24, relative to (|a| + |c|max)^2
};

template <int KS, int NP, int CT>
__global__ __launch_bounds__(kWaves * 64, 2) void assign_fast_kernel(FastArgs a) {
  constexpr int FPU = frags_per_unit(KS, NP);
  constexpr int CB = (int)chunk_bytes(KS, NP);
  constexpr int NPR = n_products(NP);
  extern __shared__ __attribute__((aligned(16))) char smem[];  // two chunk buffers
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int m = a.m, d = a.d;
  const int n_chunks = (a.n_units + kUnitsPerChunk - 1) / kUnitsPerChunk;  // the buffer is padded to whole chunks

  // LDS-DMA of chunk j into buffer j & 1: the chunk is one contiguous run of 1-KiB fragments, in
  // global memory as in LDS; wave w moves fragments w, w + 8, ...
  auto stage = [&](int j) {
    const char* src = reinterpret_cast<const char*>(a.frags) + (size_t)j * CB;
    char* dst = smem + (j & 1) * CB;
    for (int f = wave; f < CB / 1024; f += kWaves)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + f * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
  };
  stage(0);

  // this wave's points: CT column tiles of 32; fragment = dimensions 16 s + 8 half + j
  bf16x8 xs[CT][KS][NP];
  float an2[CT], an2raw[CT];  // |a - mu|^2 and |a|^2
  int pt[CT];
  bool pv[CT];
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)((int64_t)d * m * 4), 0x00020000);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    pt[ct] = (blockIdx.x * kWaves + wave) * (32 * CT) + ct * 32 + l31;
    pv[ct] = pt[ct] < m;
    int voff = pv[ct] ? (8 * half * m + pt[ct]) * 4 : 0x7ffffff0;  // out of range -> 0
    float s2 = 0.f, s2raw = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      float x[8], mk[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
        mk[j] = a.mu[16 * st + 8 * half + j];
        voff += j == 7 ? 9 * m * 4 : m * 4;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // lanes without a point and dimensions beyond d read 0 and stay 0 (mu is 0 beyond d)
        const float xc = pv[ct] ? x[j] - mk[j] : 0.f;
        __bf16 p[3];
        split3(xc, p[0], p[1], p[2]);
#pragma unroll
        for (int q = 0; q < NP; ++q) xs[ct][st][q][j] = p[q];
        s2 = fmaf(xc, xc, s2);
        s2raw = fmaf(x[j], x[j], s2raw);
      }
    }
    an2[ct] = s2 + __shfl_xor(s2, 32, 64);  // (any order: the norms only scale the bound)
    an2raw[ct] = s2raw + __shfl_xor(s2raw, 32, 64);
  }
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }
  float b1[CT], b2[CT];
  int bi[CT], bu[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    b1[ct] = b2[ct] = -INFINITY;
    bi[ct] = bu[ct] = 0;
  }
  f32x16 accA[CT], accB[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[ct][r] = -INFINITY;

  bf16x8 ar[2][NP], am[3];  // A-operand rings (see `unit`)
  constexpr int NM = CT * (1 + KS * NPR);  // MFMAs per unit
  // unit U of the chunk in `base`; `fin` = accumulators of the unit before it (uid_fin), whose values
  // go through the top-2 update between this unit's MFMAs
  auto unit = [&](auto u_c, const bf16x8* base, f32x16 (&acc)[CT], const f32x16 (&fin)[CT], int uid_fin) {
    constexpr int U = decltype(u_c)::value;
    const bf16x8* up = base + U * FPU * 64 + lane;
    float before[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) before[ct] = b1[ct];
    // A operands of the correction pass: a two-slot ring one k-step ahead that lives across the units
    // of a chunk (unit U + 1's first k-step is fetched during unit U's main pass); only unit 0 of a
    // chunk -- whose buffer is known to have landed only after the chunk barrier -- starts cold
    const bf16x8 cfrag = up[0];
    if constexpr (U == 0 || NP != 2) {
#pragma unroll
      for (int q = 0; q < NP; ++q) ar[0][q] = up[(1 + q) * 64];
    }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // epilogue slice after MFMA number mi (of NM): CT * 16 values spread over gaps [CT, NM)
    auto slice = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      if constexpr (mi >= CT) {
        constexpr int tot = CT * 16;
        constexpr int lo = ((mi - CT) * tot) / (NM - CT), hi = ((mi - CT + 1) * tot) / (NM - CT);
        static_for<lo, hi>([&](auto e_c) {
          constexpr int e = decltype(e_c)::value, ct = e / 16, r = e % 16;
          take_top2<(r & 3) + 8 * (r >> 2)>(b1[ct], b2[ct], bi[ct], fin[ct][r]);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (NP == 2) {
      // Order of accumulation = part of the error bound.  Every fp32 addition rounds relative to the
      // partial sum it produces, so the small terms go FIRST: the 2 KS correction MFMAs (c2 a1, c1 a2:
      // 32 KS additions on partial sums below 2^-7 S, S = sum |2 c_k a_k|), then the KS main MFMAs
      // (c1 a1: 16 KS additions on partial sums up to S), then -|c|^2 (3 additions):
      // (16 KS + 2 + 3) roundings of (S + |c|^2) in all, against 48 KS + 3 with the three products of
      // a k-step issued together.
      static_for<0, KS>([&](auto s_c) {
        constexpr int st = decltype(s_c)::value;
        if constexpr (st + 1 < KS) {
#pragma unroll
          for (int q = 0; q < 2; ++q) ar[(st + 1) & 1][q] = up[(1 + (st + 1) * 2 + q) * 64];
        }
        // c1 of the first two k-steps for the main pass (am: 3 slots, two k-steps ahead)
        if constexpr (st == KS - 2) am[0] = up[1 * 64];
        if constexpr (st == KS - 1) am[1] = up[(1 + 2) * 64];
        static_for<0, 2>([&](auto t_c) {
          constexpr int t = decltype(t_c)::value;  // t = 0: (c2, a1); t = 1: (c1, a2)
          static_for<0, CT>([&](auto ct_c) {
            constexpr int ct = decltype(ct_c)::value;
            if constexpr (st == 0 && t == 0) {
              acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][1], xs[ct][0][0], zero, 0, 0, 0);
            } else {
              acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[st & 1][1 - t], xs[ct][st][t], acc[ct], 0, 0, 0);
            }
            slice(std::integral_constant<int, CT * (2 * st + t) + ct>{});
          });
        });
      });
      static_for<0, KS>([&](auto s_c) {
        constexpr int st = decltype(s_c)::value;
        if constexpr (st + 2 < KS) am[(st + 2) % 3] = up[(1 + (st + 2) * 2) * 64];
        if constexpr (st == 0 && U + 1 < kUnitsPerChunk) {  // the next unit's first correction operands
          ar[0][0] = up[FPU * 64 + 1 * 64];
          ar[0][1] = up[FPU * 64 + 2 * 64];
        }
        static_for<0, CT>([&](auto ct_c) {
          constexpr int ct = decltype(ct_c)::value;
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[st % 3], xs[ct][st][0], acc[ct], 0, 0, 0);
          slice(std::integral_constant<int, CT * (2 * KS + st) + ct>{});
        });
      });
      static_for<0, CT>([&](auto ct_c) {
        constexpr int ct = decltype(ct_c)::value;
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, acc[ct], 0, 0, 0);
        slice(std::integral_constant<int, CT * 3 * KS + ct>{});
      });
    } else {
      static_for<0, CT>([&](auto ct_c) {
        constexpr int ct = decltype(ct_c)::value;
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, zero, 0, 0, 0);
        slice(std::integral_constant<int, ct>{});
      });
      static_for<0, KS>([&](auto s_c) {
        constexpr int st = decltype(s_c)::value;
        if constexpr (st + 1 < KS) {
#pragma unroll
          for (int q = 0; q < NP; ++q) ar[(st + 1) & 1][q] = up[(1 + (st + 1) * NP + q) * 64];
        }
        static_for<0, NPR>([&](auto t_c) {
          constexpr int t = decltype(t_c)::value;
          // (centroid piece, point piece), smallest first: (3,1) (1,3) (2,2) (2,1) (1,2) (1,1)
          constexpr int ca = t == 0 ? 2 : (t == 1 || t >= 4) ? 0 : 1;
          constexpr int pa = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
          static_for<0, CT>([&](auto ct_c) {
            constexpr int ct = decltype(ct_c)::value;
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[st & 1][ca], xs[ct][st][pa], acc[ct], 0, 0, 0);
            slice(std::integral_constant<int, CT * (1 + st * NPR + t) + ct>{});
          });
        });
      });
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bu[ct] = b1[ct] > before[ct] ? uid_fin : bu[ct];
  };
  using std::integral_constant;

  __syncthreads();  // chunk 0 has landed (the barrier carries the vmcnt(0) of every wave's DMA)
#pragma unroll 1
  for (int j = 0; j < n_chunks; ++j) {
    if (j + 1 < n_chunks) stage(j + 1);
    const bf16x8* base = reinterpret_cast<const bf16x8*>(smem + (j & 1) * CB);
    const int u0 = j * kUnitsPerChunk;
    unit(integral_constant<int, 0>{}, base, accA, accB, u0 - 1);
    unit(integral_constant<int, 1>{}, base, accB, accA, u0);
    unit(integral_constant<int, 2>{}, base, accA, accB, u0 + 1);
    unit(integral_constant<int, 3>{}, base, accB, accA, u0 + 2);
    // every wave is done with buffer j & 1 (the next iteration's DMA overwrites it) and chunk j+1
    // has landed
    __syncthreads();
  }
  // the last unit's values (plain code: these reads follow the MFMAs directly)
  const int uid_last = n_chunks * kUnitsPerChunk - 1;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float before = b1[ct];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = accB[ct][r];
      const float t = fminf(v, b1[ct]);
      if (v > b1[ct]) bi[ct] = (r & 3) + 8 * (r >> 2);
      b1[ct] = fmaxf(v, b1[ct]);
      b2[ct] = fmaxf(b2[ct], t);
    }
    bu[ct] = b1[ct] > before ? uid_last : bu[ct];
  }
  const float cm2 = __uint_as_float(a.cmax2_bits[0]), cm2raw = __uint_as_float(a.cmax2_bits[1]);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    int idx = bu[ct] * 32 + bi[ct] + 4 * half;
    const float o1 = __shfl_xor(b1[ct], 32, 64), o2 = __shfl_xor(b2[ct], 32, 64);
    const int oi = __shfl_xor(idx, 32, 64);
    const float B1 = fmaxf(b1[ct], o1);
    const float B2 = fmaxf(fminf(b1[ct], o1), fmaxf(b2[ct], o2));
    if (o1 > b1[ct] || (o1 == b1[ct] && oi < idx)) idx = oi;
    if (half == 0 && pv[ct]) {
      const float an = sqrtf(an2[ct]), cn = sqrtf(cm2);
      const float anr = sqrtf(an2raw[ct]), cnr = sqrtf(cm2raw);
      const float delta = 1.25f * (a.euclid ? a.eps * (an + cn) * (an + cn) + a.eps_exact * (anr + cnr) * (anr + cnr)
                                            : (a.eps + a.eps_exact) * anr * cnr);
      a.inds[pt[ct]] = idx;
      if (a.vals) a.vals[pt[ct]] = a.euclid ? B1 - an2[ct] : B1;
      // (the negated comparison also sends NaN / Inf gaps to the exact kernel)
      if (!(B1 - B2 > 2.f * delta)) a.list[atomicAdd(a.count, 1)] = pt[ct];
    }
  }
}

// ---- 2b. the same selection for BATCHED codebook-sized problems (tpq_max_sim_select) -------------
// PQ codebook training: l sub-problems, n <= 256 centroids each, d <= 64 -- the loop order of
// kmeans_split.hip: the sub-problem's centroid fragments (72 KiB at d = 64) are brought into LDS ONCE
// per block by LDS-DMA and every wave walks kSelTiles tiles of 32 points; the raw fragment of tile
// t+1 is loaded under units 0-3 of tile t and centred + split under units 4-7.  Per 32 x 32 tile:
// 2 KS correction MFMAs, KS main MFMAs, the -|c|^2 MFMA (13 at d = 64, against 25 in the six-product
// training kernel and 32 fp32 MFMAs of twice the length in the exact one); top-2 epilogue, bound and
// list exactly as in assign_fast_kernel, one list per sub-problem.
struct SelArgs {
  const float* A;        // [l][d][m]
  const bf16x8* frags;   // [l][8 units][2 KS + 1][64]
  const unsigned* cmax2_bits;  // [l][2]
  const float* mu;       // [l][256]
  int64_t* inds;         // [l][m]
  float* vals;           // optional [l][m]
  int* list;             // [l][m]
  int* count;            // [l]
  int d, m, euclid;
  float eps, eps_exact;
};
#ifndef TPQ_SEL_TILES
#define TPQ_SEL_TILES 32
#endif
constexpr int kSelTiles = TPQ_SEL_TILES;

template <int KS>
__global__ __launch_bounds__(kWaves * 64, 2) void select_resident_kernel(SelArgs a) {
  constexpr int FPU = 2 * KS + 1;
  constexpr int NM = 3 * KS + 1;  // MFMAs per unit
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16x8* fr = reinterpret_cast<const bf16x8*>(smem);          // [8][FPU][64]
  float* mu_s = reinterpret_cast<float*>(smem + 8 * FPU * 1024);     // [16 KS]
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int m = a.m, d = a.d;
  {
    const char* src = reinterpret_cast<const char*>(a.frags) + (size_t)b * 8 * FPU * 1024;
    for (int f = wave; f < 8 * FPU; f += kWaves)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + f * 1024), 16, 0, 0);
    if (threadIdx.x < 16 * KS) mu_s[threadIdx.x] = a.mu[b * 256 + threadIdx.x];
  }
  const float* __restrict__ Ab = a.A + (int64_t)b * d * m;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, (int)((int64_t)d * m * 4), 0x00020000);
  auto frag_offset = [&](int t, bool& iv, int& i) -> int {
    const int tile = blockIdx.x * kSelTiles + t;
    i = tile * (kWaves * 32) + wave * 32 + l31;
    iv = (t < kSelTiles) && (i < m);
    return iv ? (8 * half * m + i) * 4 : 0x7ffffff0;
  };
  const int row1 = m * 4, row9 = 9 * m * 4;
  float xr[KS * 8];
  bf16x8 xs[KS][2], xsn[KS][2];
  float a2c = 0.f, a2r = 0.f, a2cn = 0.f, a2rn = 0.f;  // |a - mu|^2, |a|^2 halves: current / next tile
  bool iv, ivn = false;
  int i, in_ = 0;
  __syncthreads();  // fragments (vmcnt(0) of the DMA) and mu_s are in LDS
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  // this lane's 8 centring values of k-step st (two ds_read_b128)
  auto load_mu = [&](int st, float (&mk)[8]) {
    const f32x4v lo = *reinterpret_cast<const f32x4v*>(mu_s + 16 * st + 8 * half);
    const f32x4v hi = *reinterpret_cast<const f32x4v*>(mu_s + 16 * st + 8 * half + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mk[j] = lo[j];
      mk[4 + j] = hi[j];
    }
  };
  // No predicate here: a lane without a point reads zeros and ends up with -mu in its operand -- its
  // results are never written; dimensions beyond d read 0 on both sides (mu is 0 there).  (With a
  // select on validity the compiler branched around every element: 73 exec-mask branches per tile.)
  auto split_elem = [&](auto s_c, auto j_c, bf16x8 (&dst)[KS][2], float& c2, float& r2, float muv) {
    constexpr int st = decltype(s_c)::value, j = decltype(j_c)::value;
    const float x = xr[st * 8 + j];
    const float xc = x - muv;
    __bf16 h, mm, lo;
    split3(xc, h, mm, lo);
    dst[st][0][j] = h;
    dst[st][1][j] = mm;
    // (asm: left to the compiler the two norm chains are packed into v_pk_fma_f32 with a v_mov per
    // operand pair -- 76 moves per tile)
    asm("v_fmac_f32 %0, %1, %1" : "+v"(c2) : "v"(xc));
    asm("v_fmac_f32 %0, %1, %1" : "+v"(r2) : "v"(x));
  };
  {
    int voff = frag_offset(0, iv, i);
#pragma unroll
    for (int e = 0; e < KS * 8; ++e) {
      xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
      voff += (e & 7) == 7 ? row9 : row1;
    }
    static_for<0, KS>([&](auto s_c) {
      float mk[8];
      load_mu(decltype(s_c)::value, mk);
      static_for<0, 8>([&](auto j_c) { split_elem(s_c, j_c, xs, a2c, a2r, mk[decltype(j_c)::value]); });
    });
  }
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = -INFINITY;
  // two independent top-2 chains (even / odd accumulator registers): one chain of 128 dependent
  // updates per tile left the wave waiting on its own VALU results; merged when the tile is finished
  float b1[2] = {-INFINITY, -INFINITY}, b2[2] = {-INFINITY, -INFINITY};
  int bi[2] = {0, 0}, bu[2] = {0, 0};
  float a2c_prev = 0.f, a2r_prev = 0.f;
  bool iv_prev = false;
  int i_prev = 0;
  const bf16x8* fp = fr + lane;
  bf16x8 c1k[KS], c2r[3];  // A operands (see `unit`)
  c1k[0] = fp[1 * 64];
  c2r[0] = fp[2 * 64];
  if constexpr (KS > 1) {
    c1k[1] = fp[3 * 64];
    c2r[1] = fp[4 * 64];
  }
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }
  const float cn = sqrtf(__uint_as_float(a.cmax2_bits[b * 2])), cnr = sqrtf(__uint_as_float(a.cmax2_bits[b * 2 + 1]));

  auto finish_tile = [&](bool fiv, int fi, float c2own, float r2own) {
    // merge the two chains, then the two half-waves (disjoint centroid rows of the same point)
    const int ia = bu[0] * 32 + bi[0] + 4 * half, ib = bu[1] * 32 + bi[1] + 4 * half;
    const bool tb = b1[1] > b1[0] || (b1[1] == b1[0] && ib < ia);
    int idx = tb ? ib : ia;
    const float m1 = fmaxf(b1[0], b1[1]);
    const float m2 = fmaxf(fminf(b1[0], b1[1]), fmaxf(b2[0], b2[1]));
    const float o1 = __shfl_xor(m1, 32, 64), o2 = __shfl_xor(m2, 32, 64);
    const int oi = __shfl_xor(idx, 32, 64);
    const float an2 = c2own + __shfl_xor(c2own, 32, 64), an2raw = r2own + __shfl_xor(r2own, 32, 64);
    const float B1 = fmaxf(m1, o1);
    const float B2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2));
    if (o1 > m1 || (o1 == m1 && oi < idx)) idx = oi;
    if (half == 0 && fiv) {
      const float an = sqrtf(an2), anr = sqrtf(an2raw);
      const float delta = 1.25f * (a.euclid ? a.eps * (an + cn) * (an + cn) + a.eps_exact * (anr + cnr) * (anr + cnr)
                                            : (a.eps + a.eps_exact) * anr * cnr);
      a.inds[(int64_t)b * m + fi] = idx;
      if (a.vals) a.vals[(int64_t)b * m + fi] = a.euclid ? B1 - an2 : B1;
      if (!(B1 - B2 > 2.f * delta)) a.list[(int64_t)b * m + atomicAdd(a.count + b, 1)] = fi;
    }
  };

  // unit U into `acc`; the 16 values of `fin` (unit uid_fin of the tile, or the last unit of the previous
  // tile under unit 0) go through the top-2 update between the MFMAs
  auto unit = [&](auto u_c, f32x16& acc, const f32x16& fin, int& voff_next, const bf16x8 (&xs)[KS][2],
                  bf16x8 (&xsn)[KS][2], float& c2n, float& r2n) {
    constexpr int U = decltype(u_c)::value, FU = (U + 7) & 7;
    const bf16x8* up = fp + U * FPU * 64;
    const bf16x8* upn = fp + ((U + 1) & 7) * FPU * 64;  // the next unit (unit 0 of the next tile after 7)
    const float before0 = b1[0], before1 = b1[1];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16x8 cfrag = up[0];
    float mk[8];
    if constexpr (U >= 4 && U - 4 < KS) load_mu(U - 4, mk);
    auto fill = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      if constexpr (mi >= 2) {
        constexpr int lo = ((mi - 2) * 16) / (NM - 2), hi = ((mi - 1) * 16) / (NM - 2);
        // values lo..hi-1: even registers feed chain 0, odd ones chain 1; adjacent pairs in one block
        static_for<lo, hi>([&](auto r_c) {
          constexpr int r = decltype(r_c)::value;
          if constexpr ((r & 1) == 0 && r + 1 < hi) {
            take_top2_pair<(r & 3) + 8 * (r >> 2), ((r + 1) & 3) + 8 * ((r + 1) >> 2)>(
                b1[0], b2[0], bi[0], b1[1], b2[1], bi[1], fin[r], fin[r + 1]);
          } else if constexpr ((r & 1) == 0 || r == lo) {
            take_top2<(r & 3) + 8 * (r >> 2)>(b1[r & 1], b2[r & 1], bi[r & 1], fin[r]);
          }
        });
      }
      if constexpr (U < 4) {  // next tile's raw fragment: 2 KS loads per unit
        constexpr int per = 2 * KS;
        constexpr int l0 = (mi * per) / NM, l1 = ((mi + 1) * per) / NM;
        static_for<l0, l1>([&](auto e_c) {
          constexpr int e = U * per + decltype(e_c)::value;
          xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_next, 0, 0));
          voff_next += (e & 7) == 7 ? row9 : row1;
        });
      }
      if constexpr (U >= 4 && U - 4 < KS) {  // ... centred and split under units 4..7
        constexpr int j0 = (mi * 8) / NM, j1 = ((mi + 1) * 8) / NM;
        static_for<j0, j1>([&](auto j_c) {
          split_elem(std::integral_constant<int, U - 4>{}, j_c, xsn, c2n, r2n, mk[decltype(j_c)::value]);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // small products first (see assign_fast_kernel): corrections, main, -|c|^2.  A operands: c1 of
    // every k-step stays in registers from the correction pass to the main pass (c1k), c2 runs
    // through a 3-slot ring two k-steps ahead (c2r); the first two k-steps of the NEXT unit are
    // fetched during this unit's main pass, as their registers fall free.  (One k-step ahead, and
    // the main pass reloading c1, put an LDS round trip in front of every MFMA of the main pass.)
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      if constexpr (st + 2 < KS) {
        c1k[st + 2] = up[(1 + (st + 2) * 2) * 64];
        c2r[(st + 2) % 3] = up[(2 + (st + 2) * 2) * 64];
      }
      if constexpr (st == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2r[0], xs[0][0], zero, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2r[st % 3], xs[st][0], acc, 0, 0, 0);
      }
      fill(std::integral_constant<int, 2 * st>{});
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1k[st], xs[st][1], acc, 0, 0, 0);
      fill(std::integral_constant<int, 2 * st + 1>{});
    });
    c2r[0] = upn[2 * 64];
    if constexpr (KS > 1) c2r[1] = upn[(2 + 2) * 64];
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1k[st], xs[st][0], acc, 0, 0, 0);
      if constexpr (st < 2) c1k[st] = upn[(1 + st * 2) * 64];
      fill(std::integral_constant<int, 2 * KS + st>{});
    });
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, acc, 0, 0, 0);
    fill(std::integral_constant<int, 3 * KS>{});
    bu[0] = b1[0] > before0 ? FU : bu[0];
    bu[1] = b1[1] > before1 ? FU : bu[1];
  };
  using std::integral_constant;

  bool have_prev = false;
  auto tile = [&](int t, const bf16x8 (&cur)[KS][2], bf16x8 (&nxt)[KS][2], float c2cur, float r2cur, float& c2nxt,
                  float& r2nxt) {
    int voff_next = frag_offset(t + 1, ivn, in_);
    c2nxt = 0.f;
    r2nxt = 0.f;
    unit(integral_constant<int, 0>{}, accA, accB, voff_next, cur, nxt, c2nxt, r2nxt);
    if (have_prev) finish_tile(iv_prev, i_prev, a2c_prev, a2r_prev);
    b1[0] = b1[1] = b2[0] = b2[1] = -INFINITY;
    bi[0] = bi[1] = bu[0] = bu[1] = 0;
    unit(integral_constant<int, 1>{}, accB, accA, voff_next, cur, nxt, c2nxt, r2nxt);
    unit(integral_constant<int, 2>{}, accA, accB, voff_next, cur, nxt, c2nxt, r2nxt);
    unit(integral_constant<int, 3>{}, accB, accA, voff_next, cur, nxt, c2nxt, r2nxt);
    unit(integral_constant<int, 4>{}, accA, accB, voff_next, cur, nxt, c2nxt, r2nxt);
    unit(integral_constant<int, 5>{}, accB, accA, voff_next, cur, nxt, c2nxt, r2nxt);
    unit(integral_constant<int, 6>{}, accA, accB, voff_next, cur, nxt, c2nxt, r2nxt);
    unit(integral_constant<int, 7>{}, accB, accA, voff_next, cur, nxt, c2nxt, r2nxt);
    a2c_prev = c2cur;
    a2r_prev = r2cur;
    iv_prev = iv;
    i_prev = i;
    have_prev = true;
    iv = ivn;
    i = in_;
  };
#pragma unroll 1
  for (int t = 0; t < kSelTiles; t += 2) {
    if ((blockIdx.x * kSelTiles + t) * (kWaves * 32) >= m) break;
    tile(t, xs, xsn, a2c, a2r, a2cn, a2rn);
    if (t + 1 >= kSelTiles || (blockIdx.x * kSelTiles + t + 1) * (kWaves * 32) >= m) break;
    tile(t + 1, xsn, xs, a2cn, a2rn, a2c, a2r);
  }
  if (have_prev) {  // the last unit of the last tile (plain code: these reads follow the MFMA directly)
    const float before0 = b1[0], before1 = b1[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = accB[r];
      const float t = fminf(v, b1[r & 1]);
      if (v > b1[r & 1]) bi[r & 1] = (r & 3) + 8 * (r >> 2);
      b1[r & 1] = fmaxf(v, b1[r & 1]);
      b2[r & 1] = fmaxf(b2[r & 1], t);
    }
    bu[0] = b1[0] > before0 ? 7 : bu[0];
    bu[1] = b1[1] > before1 ? 7 : bu[1];
    finish_tile(iv_prev, i_prev, a2c_prev, a2r_prev);
  }
}

// ---- 3. exact re-check: max_sim_kernel (kmeans.hip, the bit-exact fp32-MFMA kernel) over the list ----
struct Layout {
  size_t frags_off, frags_bytes, cmax_off, count_off, mu_off, list_off, keys_off, ac_off, total;
  int cap;
};
static Layout layout(int KS, int NP, int64_t m, int n, int d = 0) {
  Layout L;
  const int64_t units = (n + 31) / 32;
  const int64_t chunks = (units + kUnitsPerChunk - 1) / kUnitsPerChunk;
  L.frags_off = 0;
  L.frags_bytes = (size_t)chunks * chunk_bytes(KS, NP);
  L.cmax_off = L.frags_bytes;
  L.count_off = L.cmax_off + 256;
  L.mu_off = L.count_off + 256;
  L.list_off = L.mu_off + 1024;
  L.keys_off = (L.list_off + (size_t)m * 4 + 255) / 256 * 256;  // [m] u64: the split re-check's (value, index) keys
  // compact copy of the re-checked points' columns: a quarter of the points (at least 8192) -- the
  // re-check share is a few per cent; more than that overflows into the gathering launch
  L.cap = (int)(m < 8192 ? m : (m / 4 > 8192 ? m / 4 : 8192));
  L.ac_off = (L.keys_off + (size_t)m * 8 + 255) / 256 * 256;
  L.total = L.ac_off + (size_t)L.cap * d * 4;
  return L;
}

template <int KS, int NP, int CT>
static int run(const float* A, const float* B, float* vals, int64_t* inds, int d, int m, int n, int euclid, char* ws,
               hipStream_t st) {
  const Layout L = layout(KS, NP, m, n, d);
  bf16x8* frags = reinterpret_cast<bf16x8*>(ws + L.frags_off);
  unsigned* cmax = reinterpret_cast<unsigned*>(ws + L.cmax_off);
  int* count = reinterpret_cast<int*>(ws + L.count_off);
  int* list = reinterpret_cast<int*>(ws + L.list_off);
  const int units = (n + 31) / 32;
  const int units_padded = (units + kUnitsPerChunk - 1) / kUnitsPerChunk * kUnitsPerChunk;
  float* mu = reinterpret_cast<float*>(ws + L.mu_off);
  int rc = check_hip(hipMemsetAsync(ws + L.cmax_off, 0, 512 + 1024, st), "coarse_assign memset");
  if (rc) return rc;
  hipLaunchKernelGGL(assign_mean_kernel, dim3(d), dim3(256), 0, st, B, mu, n, euclid);
  TPQ_LAUNCH_CHECK("assign_mean_kernel");
  // padding units: rows beyond n get -3e38 norms and zero pieces from the kernel itself
  hipLaunchKernelGGL((assign_prep_kernel<KS, NP>), dim3(units_padded), dim3(64), 0, st, B, frags, cmax, mu, d, n,
                     euclid);
  TPQ_LAUNCH_CHECK("assign_prep_kernel");
  const size_t lds = 2 * chunk_bytes(KS, NP);
  auto kernel = assign_fast_kernel<KS, NP, CT>;
  rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                 "assign_fast_kernel attr");
  if (rc) return rc;
  // roundings of the full-size partial sum (see the accumulation order in the kernel)
  const int terms = NP == 2 ? KS * 16 + 2 + 3 : KS * 16 * n_products(NP) + 3;
  // dropped products: NP = 2: c2 a2 + r_c a + c r_a <= 3 x 2^-16 |a_k c_k| (1 % slack for the second-order
  // terms); NP = 3: c2 a3 + c3 a2 + c3 a3 <= 2^-23 |a_k c_k|
  const float eps_prod = NP == 2 ? 3.03f / 65536.0f : 1.01f / 8388608.0f;
  // (+ 8: the shift by mu rounds both operands once, 2^-23 (|a - mu| + |c - mu|)^2, and slack)
  FastArgs fa{A, frags, cmax, mu, inds, vals, list, count, d, m, units_padded, euclid,
              eps_prod + (float)(terms + 8) / 8388608.0f, (float)(d + 4) / 16777216.0f};
  const int per_block = kWaves * 32 * CT;
  hipLaunchKernelGGL(kernel, dim3((m + per_block - 1) / per_block), dim3(kWaves * 64), lds, st, fa);
  TPQ_LAUNCH_CHECK("assign_fast_kernel");
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + L.keys_off);
  rc = check_hip(hipMemsetAsync(keys, 0, (size_t)m * 8, st), "coarse_assign keys memset");
  if (rc) return rc;
  return launch_max_sim_list(A, B, vals, inds, 1, d, m, n, euclid, list, count, keys,
                             reinterpret_cast<float*>(ws + L.ac_off), L.cap, st);
}

struct SelLayout {
  size_t frags_bytes, cmax_off, count_off, mu_off, list_off, total;
};
static SelLayout sel_layout(int KS, int l, int64_t m) {
  SelLayout L;
  L.frags_bytes = (size_t)l * 8 * (2 * KS + 1) * 1024;
  L.cmax_off = L.frags_bytes;                          // [l][2] u32
  L.count_off = L.cmax_off + (size_t)l * 8;            // [l] i32
  L.mu_off = (L.count_off + (size_t)l * 4 + 255) / 256 * 256;  // [l][256] f32
  L.list_off = L.mu_off + (size_t)l * 1024;            // [l][m] i32
  L.total = L.list_off + (size_t)l * m * 4;
  return L;
}

template <int KS>
static int run_select(const float* A, const float* B, float* vals, int64_t* inds, int l, int d, int m, int n,
                      int euclid, char* ws, hipStream_t st) {
  const SelLayout L = sel_layout(KS, l, m);
  bf16x8* frags = reinterpret_cast<bf16x8*>(ws);
  unsigned* cmax = reinterpret_cast<unsigned*>(ws + L.cmax_off);
  int* count = reinterpret_cast<int*>(ws + L.count_off);
  float* mu = reinterpret_cast<float*>(ws + L.mu_off);
  int* list = reinterpret_cast<int*>(ws + L.list_off);
  int rc = check_hip(hipMemsetAsync(ws + L.cmax_off, 0, L.list_off - L.cmax_off, st), "max_sim_select memset");
  if (rc) return rc;
  hipLaunchKernelGGL(assign_mean_kernel, dim3(d, l), dim3(256), 0, st, B, mu, n, euclid);
  TPQ_LAUNCH_CHECK("assign_mean_kernel");
  hipLaunchKernelGGL((assign_prep_kernel<KS, 2>), dim3(8, l), dim3(64), 0, st, B, frags, cmax, mu, d, n, euclid);
  TPQ_LAUNCH_CHECK("assign_prep_kernel");
  const size_t lds = (size_t)8 * (2 * KS + 1) * 1024 + 16 * KS * 4;
  auto kernel = select_resident_kernel<KS>;
  rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                 "select_resident_kernel attr");
  if (rc) return rc;
  const int terms = KS * 16 + 2 + 3;
  SelArgs sa{A, frags, cmax, mu, inds, vals, list, count, d, m, euclid,
             3.03f / 65536.0f + (float)(terms + 8) / 8388608.0f, (float)(d + 4) / 16777216.0f};
  const int per_block = kWaves * 32 * kSelTiles;
  hipLaunchKernelGGL(kernel, dim3((m + per_block - 1) / per_block, l), dim3(kWaves * 64), lds, st, sa);
  TPQ_LAUNCH_CHECK("select_resident_kernel");
  return launch_max_sim_list(A, B, vals, inds, l, d, m, n, euclid, list, count, nullptr, nullptr, 0, st);
}

}  // namespace afast
}  // namespace tpq

using namespace tpq;

static int sel_ks(int d) { return d <= 16 ? 1 : (d <= 32 ? 2 : (d <= 48 ? 3 : 4)); }

extern "C" int tpq_max_sim_select_supported(int l, int d, int64_t m, int n) {
  return (l >= 1 && l <= 65535 && d >= 1 && d <= 64 && n >= 1 && n <= 256 && m >= 0 && m < (1LL << 31) &&
          (int64_t)sel_ks(d) * 16 * m * 4 <= 0x7fffffffLL)
             ? 1
             : 0;
}

extern "C" size_t tpq_max_sim_select_workspace_bytes(int l, int d, int64_t m, int n) {
  if (!tpq_max_sim_select_supported(l, d, m, n)) return 0;
  return afast::sel_layout(sel_ks(d), l, m).total;
}

extern "C" int tpq_max_sim_select(const float* A, const float* B, float* vals, int64_t* inds, int l, int d,
                                  int64_t m, int n, int metric, void* workspace, size_t workspace_bytes,
                                  tpq_stream_t stream) {
  TPQ_REQUIRE(A && B && inds, "max_sim_select: null pointer");
  TPQ_REQUIRE(metric == TPQ_METRIC_NEG_SQ_L2 || metric == TPQ_METRIC_INNER, "max_sim_select: bad metric %d", metric);
  if (!tpq_max_sim_select_supported(l, d, m, n)) {
    set_error("max_sim_select: shape l=%d d=%d m=%lld n=%d not supported (d <= 64, n <= 256); use tpq_max_sim", l, d,
              (long long)m, n);
    return TPQ_ERR_UNSUPPORTED;
  }
  if (m == 0) return TPQ_OK;
  const size_t need = tpq_max_sim_select_workspace_bytes(l, d, m, n);
  TPQ_REQUIRE(workspace && workspace_bytes >= need, "max_sim_select: workspace of %zu bytes needed", need);
  const int euclid = metric == TPQ_METRIC_NEG_SQ_L2 ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  switch (sel_ks(d)) {
    case 1: return afast::run_select<1>(A, B, vals, inds, l, d, (int)m, n, euclid, ws, st);
    case 2: return afast::run_select<2>(A, B, vals, inds, l, d, (int)m, n, euclid, ws, st);
    case 3: return afast::run_select<3>(A, B, vals, inds, l, d, (int)m, n, euclid, ws, st);
    default: return afast::run_select<4>(A, B, vals, inds, l, d, (int)m, n, euclid, ws, st);
  }
}

#ifndef TPQ_AF_NP
#define TPQ_AF_NP 2
#endif
#ifndef TPQ_AF_CT
#define TPQ_AF_CT 2
#endif

static int af_ks(int d) { return d <= 32 ? 2 : (d <= 64 ? 4 : 8); }

extern "C" int tpq_coarse_assign_supported(int d, int64_t m, int n) {
  if (d > 128) return m == 0 ? (d <= 1024 && n >= 1) : lloyd_wide_supported(d, m, n);  // the GEMM-shaped cascade
  return (d >= 1 && d <= 128 && n >= 1 && n <= (1 << 24) && m >= 0 && m < (1LL << 31) &&
          (int64_t)af_ks(d) * 16 * m * 4 <= 0x7fffffffLL)
             ? 1
             : 0;
}

// d > 128: the cascade's dozen launches and the preparation of the points cost ~0.2 ms + a pass over the
// data; below this many multiply-adds the fp32 kernel is done sooner (same labels either way).
// (route == TPQ_ASSIGN_ROUTE_CASCADE: the caller asks for the cascade whatever the size -- tests, tuning)
static bool wide_cascade_pays(int d, int64_t m, int n, int route) {
  if (route == TPQ_ASSIGN_ROUTE_CASCADE) return true;
  return (double)m * (double)n * (double)d >= 8589934592.0;
}

// workspace = [the two-piece bf16 selection's layout][the cascade's layout]: either path may run (the
// cascade takes euclidean problems, the selection inner products), the diagnostics word stays where it was
static size_t af_old_total(int d, int64_t m, int n) {
  if (d > 128) return 256;  // wide vectors: the diagnostics word alone
  return (afast::layout(af_ks(d), TPQ_AF_NP, m, n, d).total + 255) / 256 * 256;
}
static bool route_ok(int route) { return route == TPQ_ASSIGN_ROUTE_AUTO || route == TPQ_ASSIGN_ROUTE_CASCADE; }

extern "C" size_t tpq_coarse_assign_route_workspace_bytes(int d, int64_t m, int n, int route) {
  if (!tpq_coarse_assign_supported(d, m, n) || !route_ok(route)) return 0;
  if (d > 128) {  // [diagnostics word][the cascade's layout, or -- small problems -- the fp32 kernel's maxima]
    const size_t cascade = m > 0 ? lloyd_assign_workspace_bytes(d, m, n) : 0, plain = ((size_t)m * 4 + 255) / 256 * 256;
    return 256 + (cascade > plain ? cascade : plain);
  }
  return af_old_total(d, m, n) +
         (m > 0 && lloyd_assign_supported(d, m, n, route) ? lloyd_assign_workspace_bytes(d, m, n) : 0);
}
extern "C" size_t tpq_coarse_assign_workspace_bytes(int d, int64_t m, int n) {
  return tpq_coarse_assign_route_workspace_bytes(d, m, n, TPQ_ASSIGN_ROUTE_AUTO);
}

// diagnostics: byte offset, inside the workspace, of the int32 number of points the last call sent
// to the exact re-check
extern "C" size_t tpq_coarse_assign_count_offset(int d, int64_t m, int n) {
  if (!tpq_coarse_assign_supported(d, m, n) || d > 128) return 0;
  return afast::layout(af_ks(d), TPQ_AF_NP, m, n).count_off;
}

extern "C" int tpq_coarse_assign_route(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m,
                                       int n, int metric, int route, void* workspace, size_t workspace_bytes,
                                       tpq_stream_t stream) {
  TPQ_REQUIRE(A && B && inds, "coarse_assign: null pointer");
  TPQ_REQUIRE(metric == TPQ_METRIC_NEG_SQ_L2 || metric == TPQ_METRIC_INNER, "coarse_assign: bad metric %d", metric);
  TPQ_REQUIRE(route_ok(route), "coarse_assign: bad route %d", route);
  if (!tpq_coarse_assign_supported(d, m, n)) {
    set_error("coarse_assign: shape d=%d m=%lld n=%d not supported (d <= 1024; d <= 128: padded slice < 2 GiB; "
              "d > 128: m < 2^28, n <= 2^22); use tpq_max_sim",
              d, (long long)m, n);
    return TPQ_ERR_UNSUPPORTED;
  }
  if (m == 0) return TPQ_OK;
  const size_t need = tpq_coarse_assign_route_workspace_bytes(d, m, n, route);
  TPQ_REQUIRE(workspace && workspace_bytes >= need, "coarse_assign: workspace of %zu bytes needed", need);
  const int euclid = metric == TPQ_METRIC_NEG_SQ_L2 ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  if (d > 128 && !wide_cascade_pays(d, m, n, route)) {  // a small problem: the fp32 kernel itself
    check_hip(hipMemsetAsync(ws, 0, 4, st), "coarse_assign memset");
    return tpq_max_sim(A, B, vals ? vals : reinterpret_cast<float*>(ws + 256), inds, 1, d, (int)m, n, metric, stream);
  }
  if (d > 128 || (euclid && lloyd_assign_supported(d, m, n, route))) {  // the fp16 cascade (lloyd.hip)
    char* cws = ws + af_old_total(d, m, n);
    int rc = lloyd_assign(A, B, vals, inds, d, m, n, euclid, cws, st);
    if (rc) return rc;
    // diagnostics: the number of exactly re-checked points where tpq_coarse_assign_count_offset points
    return check_hip(hipMemcpyAsync(ws + (d > 128 ? 0 : afast::layout(af_ks(d), TPQ_AF_NP, m, n).count_off),
                                    cws + lloyd_assign_count_offset(d, m, n), 4, hipMemcpyDeviceToDevice, st),
                     "coarse_assign count copy");
  }
  switch (af_ks(d)) {
    case 2: return afast::run<2, TPQ_AF_NP, TPQ_AF_CT>(A, B, vals, inds, d, (int)m, n, euclid, ws, st);
    case 4: return afast::run<4, TPQ_AF_NP, TPQ_AF_CT>(A, B, vals, inds, d, (int)m, n, euclid, ws, st);
    default: return afast::run<8, TPQ_AF_NP, TPQ_AF_CT>(A, B, vals, inds, d, (int)m, n, euclid, ws, st);
  }
}

extern "C" int tpq_coarse_assign(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n,
                                 int metric, void* workspace, size_t workspace_bytes, tpq_stream_t stream) {
  return tpq_coarse_assign_route(A, B, vals, inds, d, m, n, metric, TPQ_ASSIGN_ROUTE_AUTO, workspace, workspace_bytes,
                                 stream);
}
