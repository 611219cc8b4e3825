// K-means assign on the bf16 matrix cores with fp32-level accuracy (training path of
// MultiKMeans.fit; the bit-exact fp32-MFMA kernels of kmeans.hip stay the encode / predict path).
//
// tpq_max_sim_split computes the same (max, arg-max) of 2 a.b - |a|^2 - |b|^2 (or a.b) as
// tpq_max_sim (replaces max_sim_tn, torchpq/kernels/cuda/max_sim.cu:182-309, as called from the
// Lloyd loop torchpq/clustering/MultiKMeans.py:415-453), but the contraction runs on
// v_mfma_f32_32x32x16_bf16 at 16x the fp32-MFMA rate:
//   * every fp32 value is split EXACTLY into three bf16 pieces, x = x1 + x2 + x3 (x1 = bf16(x),
//     x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 3 x 8 significant bits, round to nearest even);
//   * a.b = sum_{i,j} a_i b_j; every product of two pieces is exact in fp32; the six products of
//     order i + j <= 4 are accumulated (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2), the three dropped ones
//     (a2b3, a3b2, a3b3) are bounded by (2^-24 + 2^-24 + 2^-32) |a_k| |b_k| per term -- the size of
//     ONE fp32 rounding of the product, i.e. below the rounding error the fp32 fma chain itself
//     commits (d roundings).  tests/test_gpu_kernels.py checks both kernels against float64.
//   * 6 bf16 MFMAs (32 cycles each) replace 8 fp32 MFMAs (64 cycles each) per 16 k: 2.67x fewer
//     matrix-pipe cycles, and the bf16 MFMA runs in the matrix core proper, so the arg-max
//     epilogue, the splitting and the loads issue UNDER it (the fp32 MFMA executes on the SIMD's
//     fp32 ALUs: VALU work next to it adds to the time).
// Not bit-identical to the fp32 kernel (different rounding points), so arg-max ties and near-ties
// (gap below ~1e-6 relative) can resolve differently: labels feed centroid averages in training,
// never stored codes.
//
// Structure: centroids are the MFMA rows, points the columns (each lane owns ONE point: the
// arg-max over centroids is an in-lane reduction over accumulator registers, as in kmeans.hip).
// A block of 8 waves stages the <= 256 centroids of sub-problem b ONCE -- scaled by 2 (exact),
// split, laid out in LDS in MFMA-fragment order [unit][k-step][piece][lane] x 16 B, so that every A
// operand is one conflict-free ds_read_b128 at lane*16 + constant -- plus -|c|^2 per centroid,
// which seeds the accumulator: after the chain acc = 2 a.c - |c|^2 and the epilogue is a bare
// compare + two selects per value.  Each wave then walks kSpTiles tiles of 32 points; the raw fp32
// fragment of tile t+1 is loaded under units 0-3 of tile t and split under units 4-7.
#include <type_traits>

#include "common.h"

namespace tpq {
namespace split {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

// (best, besti) <- (val, CL) if val > best.  CL is an inline constant (0..64): see ms_take in
// kmeans.hip.  Inputs are accumulator registers of the PREVIOUS unit, whose last MFMA retired at
// least one full MFMA issue slot earlier (the callers place no slice before the second MFMA of
// the running unit), so no MFMA->VALU wait states are owed inside the asm.
template <int CL>
__device__ __forceinline__ void take(float& best, int& besti, float val) {
  static_assert(CL >= 0 && CL <= 64, "inline constant");
  asm volatile("v_cmp_ngt_f32 vcc, %2, %0\n\tv_cndmask_b32 %0, %2, %0, vcc\n\tv_cndmask_b32 %1, %3, %1, vcc"
               : "+v"(best), "+v"(besti)
               : "v"(val), "n"(CL)
               : "vcc");
}

// two values in one block (separate asm statements get a wait state between them)
template <int CL0, int CL1>
__device__ __forceinline__ void take2(float& best, int& besti, float v0, float v1) {
  static_assert(CL0 >= 0 && CL0 <= 64 && CL1 >= 0 && CL1 <= 64, "inline constants");
  asm volatile(
      "v_cmp_ngt_f32 vcc, %2, %0\n\tv_cndmask_b32 %0, %2, %0, vcc\n\tv_cndmask_b32 %1, %4, %1, vcc\n\t"
      "v_cmp_ngt_f32 vcc, %3, %0\n\tv_cndmask_b32 %0, %3, %0, vcc\n\tv_cndmask_b32 %1, %5, %1, vcc"
      : "+v"(best), "+v"(besti)
      : "v"(v0), "v"(v1), "n"(CL0), "n"(CL1)
      : "vcc");
}

// x -> (x1, x2, x3), exact: x == x1 + x2 + x3
__device__ __forceinline__ void split3(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

#ifndef TPQ_SP_TILES
#define TPQ_SP_TILES 32
#endif
constexpr int kSpTiles = TPQ_SP_TILES;  // 32-point tiles per wave (a block covers 8 x 32 x kSpTiles points)
constexpr int kSpWaves = 8;
#ifndef TPQ_SP_EXP
#define TPQ_SP_EXP 0  // knock-outs (tools/build_variant.sh): 2 no in-loop split, 4 no loads
#endif

constexpr size_t split_lds_bytes(int KS) { return (size_t)(8 * KS * 3 + 8) * 64 * 16; }

// KS = k-steps of 16 dimensions (d <= 16 KS)
template <int KS, bool euclidean>
__global__ __launch_bounds__(kSpWaves * 64, 2) void max_sim_split_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ vals,
    int64_t* __restrict__ inds, int d, int m, int n_total, int c0, int first) {
  const int n = (n_total - c0) < 256 ? (n_total - c0) : 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* cpl = reinterpret_cast<bf16x8*>(smem);  // [8 units][KS][3 pieces][64 lanes]
  // -|c|^2 as one more A fragment per unit: row c holds its three bf16 pieces at k = 0, 1, 2 (lanes
  // of k-group 0), zero elsewhere; against a B fragment of ones at k = 0, 1, 2 the MFMA adds exactly
  // -|c|^2 to every column -- the accumulator chain starts from the literal 0 and needs no seed
  // registers (seeding from LDS put 4 ds_read_b128 + their latency at the head of every unit)
  bf16x8* cnf = reinterpret_cast<bf16x8*>(smem) + 8 * KS * 3 * 64;  // [8 units][64 lanes]
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const float* __restrict__ Ab = A + (int64_t)b * d * m;
  const float* __restrict__ Bb = B + (int64_t)b * d * n_total + c0;

  // ---- stage the centroids: item = (centroid c, group g of 8 dimensions) ----------------------
  for (int e = threadIdx.x; e < 256 * 2 * KS; e += kSpWaves * 64) {
    const int c = e & 255, g = e >> 8;
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * g + j;
      float x = (k < d && c < n) ? Bb[(int64_t)k * n_total + c] : 0.f;
      if (euclidean) x *= 2.f;  // exact; the chain then yields 2 a.c
      __bf16 h, mm, lo;
      split3(x, h, mm, lo);
      p1[j] = h;
      p2[j] = mm;
      p3[j] = lo;
    }
    const int slot = (((c >> 5) * KS + (g >> 1)) * 3) * 64 + (g & 1) * 32 + (c & 31);
    cpl[slot] = p1;
    cpl[slot + 64] = p2;
    cpl[slot + 128] = p3;
  }
  {
    // |c|^2: thread (c, hf) sums every other dimension (independent loads), the pair meets in LDS.
    // Padding columns get -3e38: they can never win.
    float* part = reinterpret_cast<float*>(cnf);  // scratch: [2][256] floats inside the 8 KiB of cnf
    const int c = threadIdx.x & 255, hf = threadIdx.x >> 8;
    float s0 = 0.f, s1 = 0.f;
    if (euclidean && c < n) {
      int k = hf;
      for (; k + 2 < d; k += 4) {
        const float x0 = Bb[(int64_t)k * n_total + c], x1 = Bb[(int64_t)(k + 2) * n_total + c];
        s0 = fmaf(x0, x0, s0);
        s1 = fmaf(x1, x1, s1);
      }
      if (k < d) {
        const float x0 = Bb[(int64_t)k * n_total + c];
        s0 = fmaf(x0, x0, s0);
      }
    }
    part[hf * 256 + c] = s0 + s1;
    __syncthreads();
    float nrm = 0.f;
    if (threadIdx.x < 256) nrm = c < n ? -(part[c] + part[256 + c]) : -3.0e38f;
    __syncthreads();
    // fragment [unit c/32][lane (k-group) * 32 + c%32]: k-group 0 carries the pieces, k-group 1 zeros
    bf16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x < 256) {
      __bf16 h, mm, lo;
      split3(nrm, h, mm, lo);
      f[0] = h;
      f[1] = mm;
      f[2] = lo;
    }
    cnf[(c >> 5) * 64 + hf * 32 + (c & 31)] = f;  // threads 256..511 write the zero k-group
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(Ab), 0, (int)((int64_t)d * m * 4), 0x00020000);
  // this lane's fragment of tile t: point i, dimensions 16 s + 8 half + j (s < KS, j < 8)
  auto frag_offset = [&](int t, bool& iv, int& i) -> int {
    const int tile = blockIdx.x * kSpTiles + t;
    i = tile * (kSpWaves * 32) + wave * 32 + l31;
    iv = (t < kSpTiles) && (i < m);
    return iv ? (8 * half * m + i) * 4 : 0x7ffffff0;  // out of range -> the loads return 0
  };
  const int row1 = m * 4, row9 = 9 * m * 4;  // next dimension / first dimension of the next k-step
  float xr[KS * 8];       // raw fp32 fragment in flight (next tile)
  bf16x8 xs[KS][3];       // current tile's B operands: [k-step][piece]
  bf16x8 xsn[KS][3];      // next tile's, filled under units 4..7
  float a2 = 0.f, a2n = 0.f;  // this lane's half of |a|^2 (current / next tile)
  bool iv, ivn = false;
  int i, in_ = 0;
  // split element j of k-step s of the raw fragment into dst (and fold it into |a|^2)
  auto split_elem = [&](auto s_c, auto j_c, bf16x8 (&dst)[KS][3], float& a2acc) {
    constexpr int s = decltype(s_c)::value, j = decltype(j_c)::value;
    const float x = xr[s * 8 + j];
    __bf16 h, mm, lo;
    split3(x, h, mm, lo);
    dst[s][0][j] = h;
    dst[s][1][j] = mm;
    dst[s][2][j] = lo;
    if (euclidean) a2acc = fmaf(x, x, a2acc);
  };
  {
    int voff = frag_offset(0, iv, i);
#pragma unroll
    for (int e = 0; e < KS * 8; ++e) {
      xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
      voff += (e & 7) == 7 ? row9 : row1;
    }
    static_for<0, KS>([&](auto s_c) {
      static_for<0, 8>([&](auto j_c) { split_elem(s_c, j_c, xs, a2); });
    });
  }

  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = -INFINITY;
  float best = -INFINITY;
  int besti = 0, bestu = 0;
  float a2_prev = 0.f;
  bool iv_prev = false;
  int i_prev = 0;
  const bf16x8* cp = cpl + lane;
  bf16x8 ar[2][3];  // A operands of k-step g (ring slot g & 1), fetched one k-step ahead
#pragma unroll
  for (int p = 0; p < 3; ++p) ar[0][p] = cp[p * 64];
  bf16x8 cfrag = cnf[lane];  // -|c|^2 fragment of the coming unit
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};  // B fragment of ones at k = 0, 1, 2
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }

  auto finish_tile = [&](bool fiv, int fi, float a2own) {
    besti += 32 * bestu + 4 * half;
    const float ov = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(besti, 32, 64);
    const float a2o = __shfl_xor(a2own, 32, 64);
    if (ov > best || (ov == best && oi < besti)) {
      best = ov;
      besti = oi;
    }
    if (half == 0 && fiv) {
      if (euclidean) best -= (a2own + a2o);
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

  constexpr int NM = KS * 6;  // MFMAs per unit
  // unit U of the current tile into `acc`; `fin` = accumulator of the unit before it, whose 16
  // values go through the arg-max between this unit's MFMAs (from the second MFMA on)
  auto unit = [&](auto u_c, f32x16& acc, const f32x16& fin, int& voff_next, const bf16x8 (&xs)[KS][3],
                  bf16x8 (&xsn)[KS][3], float& a2n) {
    constexpr int U = decltype(u_c)::value, FU = (U + 7) & 7;
    const float best_before = best;
    {
      // (inner product: the fragment is 0 for real centroids and -3e38 for padding columns)
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, zero, 0, 0, 0);
      cfrag = cnf[((U + 1) & 7) * 64 + lane];  // the next unit's, a whole unit ahead
    }
    static_for<0, KS>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int g = U * KS + s;  // k-step counter over the tile: ring slot g & 1 (8 KS is even)
      {
        // A operands one k-step ahead, across unit and tile boundaries (after unit 7 comes unit 0)
        constexpr int gn = (g + 1) % (8 * KS);
#pragma unroll
        for (int p = 0; p < 3; ++p) ar[(g + 1) & 1][p] = cp[(gn * 3 + p) * 64];
      }
      const bf16x8 a1 = ar[g & 1][0], a2p = ar[g & 1][1], a3 = ar[g & 1][2];
      static_for<0, 6>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        constexpr int mi = s * 6 + t;  // MFMA index within the unit
        // smallest products first: a3b1, a1b3, a2b2, a2b1, a1b2, a1b1 -- one chain per unit: the second
        // wave of the SIMD fills the matrix pipe between dependent MFMAs (two chains per unit, even / odd
        // MFMAs, measured slower: 10.2 vs 9.8 ms at C5)
        const bf16x8 aop = t == 0 ? a3 : (t == 1 || t >= 4) ? a1 : a2p;
        const bf16x8 bop = t == 0 ? xs[s][0] : t == 1 ? xs[s][2] : t == 2 ? xs[s][1] : t == 3 ? xs[s][0]
                                                                                    : t == 4 ? xs[s][1] : xs[s][0];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aop, bop, acc, 0, 0, 0);
        // arg-max slices of the previous unit: 16 values over MFMA gaps [2, NM) (or all after the
        // last MFMA when the unit is too short)
        if constexpr (NM > 2 && mi >= 2) {
          constexpr int lo = ((mi - 2) * 16) / (NM - 2), hi = ((mi - 1) * 16) / (NM - 2);
          constexpr int np = (hi - lo) / 2;
          static_for<0, np>([&](auto q_c) {
            constexpr int r = lo + 2 * decltype(q_c)::value;
            take2<(r & 3) + 8 * (r >> 2), ((r + 1) & 3) + 8 * ((r + 1) >> 2)>(best, besti, fin[r], fin[r + 1]);
          });
          if constexpr ((hi - lo) & 1) {
            constexpr int r = hi - 1;
            take<(r & 3) + 8 * (r >> 2)>(best, besti, fin[r]);
          }
        }
        // next tile's raw fragment: 2 KS loads per unit under units 0..3
        if constexpr (U < 4) {
          constexpr int per = 2 * KS;
          constexpr int l0 = (mi * per) / NM, l1 = ((mi + 1) * per) / NM;
          static_for<l0, l1>([&](auto e_c) {
            constexpr int e = U * per + decltype(e_c)::value;
            if (!(TPQ_SP_EXP & 4)) xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_next, 0, 0));
            voff_next += (e & 7) == 7 ? row9 : row1;
          });
        }
        // ... and split under units 4..7 (k-step U - 4): 8 elements over the unit's MFMA gaps
        // (spreading load + split evenly over all 8 units, loads one tile further ahead, measured
        // slower: 9.9 vs 9.55 ms at C5 -- 256 registers instead of 230)
        if constexpr (U >= 4 && U - 4 < KS) {
          constexpr int j0 = (mi * 8) / NM, j1 = ((mi + 1) * 8) / NM;
          if (!(TPQ_SP_EXP & 2))
            static_for<j0, j1>([&](auto j_c) { split_elem(std::integral_constant<int, U - 4>{}, j_c, xsn, a2n); });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    bestu = best > best_before ? FU : bestu;
  };
  using std::integral_constant;

  bool have_prev = false;
  // one tile: `cur` holds its B operands, `nxt` receives the next tile's (the two buffers swap
  // roles from tile to tile: the loop below is unrolled by two instead of copying 12 KS registers)
  auto tile = [&](int t, const bf16x8 (&cur)[KS][3], bf16x8 (&nxt)[KS][3], float a2cur, float& a2nxt) {
    int voff_next = frag_offset(t + 1, ivn, in_);
    a2nxt = 0.f;
    // unit 0 carries the arg-max of the LAST unit of tile t-1 (accB = -inf before the first tile)
    unit(integral_constant<int, 0>{}, accA, accB, voff_next, cur, nxt, a2nxt);
    if (have_prev) finish_tile(iv_prev, i_prev, a2_prev);
    best = -INFINITY;
    besti = 0;
    bestu = 0;
    unit(integral_constant<int, 1>{}, accB, accA, voff_next, cur, nxt, a2nxt);
    unit(integral_constant<int, 2>{}, accA, accB, voff_next, cur, nxt, a2nxt);
    unit(integral_constant<int, 3>{}, accB, accA, voff_next, cur, nxt, a2nxt);
    unit(integral_constant<int, 4>{}, accA, accB, voff_next, cur, nxt, a2nxt);
    unit(integral_constant<int, 5>{}, accB, accA, voff_next, cur, nxt, a2nxt);
    unit(integral_constant<int, 6>{}, accA, accB, voff_next, cur, nxt, a2nxt);
    unit(integral_constant<int, 7>{}, accB, accA, voff_next, cur, nxt, a2nxt);
    a2_prev = a2cur;
    iv_prev = iv;
    i_prev = i;
    have_prev = true;
    iv = ivn;
    i = in_;
  };
#pragma unroll 1
  for (int t = 0; t < kSpTiles; t += 2) {
    if ((blockIdx.x * kSpTiles + t) * (kSpWaves * 32) >= m) break;
    tile(t, xs, xsn, a2, a2n);
    if (t + 1 >= kSpTiles || (blockIdx.x * kSpTiles + t + 1) * (kSpWaves * 32) >= m) break;
    tile(t + 1, xsn, xs, a2n, a2);
  }
  if (have_prev) {  // arg-max of the last unit of the last tile
    const float best_before = best;
    static_for<0, 16>([&](auto r_c) {
      constexpr int r = decltype(r_c)::value;
      // (plain code: these reads follow the MFMA directly and need the compiler's wait states)
      const float v = accB[r];
      if (v > best) {
        best = v;
        besti = (r & 3) + 8 * (r >> 2);
      }
    });
    bestu = best > best_before ? 7 : bestu;
    finish_tile(iv_prev, i_prev, a2_prev);
  }
}

template <int KS>
static int launch_split(const float* A, const float* B, float* vals, int64_t* inds, int l, int d, int m,
                        int n, int euclid, hipStream_t st) {
  const size_t lds = split_lds_bytes(KS);
  const int per_block = kSpWaves * 32 * kSpTiles;
  const dim3 grid((m + per_block - 1) / per_block, l);
  auto go = [&](auto kernel) -> int {
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "max_sim_split_kernel attr");
    if (rc) return rc;
    for (int c0 = 0; c0 < n; c0 += 256) {
      hipLaunchKernelGGL(kernel, grid, dim3(kSpWaves * 64), lds, st, A, B, vals, inds, d, m, n, c0,
                         c0 == 0 ? 1 : 0);
      TPQ_LAUNCH_CHECK("max_sim_split_kernel");
    }
    return TPQ_OK;
  };
  return euclid ? go(max_sim_split_kernel<KS, true>) : go(max_sim_split_kernel<KS, false>);
}

}  // namespace split
}  // namespace tpq

using namespace tpq;

extern "C" int tpq_max_sim_split_supported(int d, int64_t m, int n) {
  // (32-bit buffer offsets: the padded slice, 16 ceil(d/16) rows of m floats, must stay below 2 GiB)
  return (d >= 1 && d <= 64 && n >= 1 && m >= 0 && (int64_t)((d + 15) / 16) * 16 * m * 4 <= 0x7fffffffLL) ? 1 : 0;
}

extern "C" int tpq_max_sim_split(const float* A, const float* B, float* vals, int64_t* inds, int l, int d,
                                 int m, int n, int metric, tpq_stream_t stream) {
  TPQ_REQUIRE(A && B && vals && inds, "max_sim_split: null pointer");
  TPQ_REQUIRE(l >= 1 && d >= 1 && m >= 0 && n >= 1, "max_sim_split: bad shape l=%d d=%d m=%d n=%d", l, d, m, n);
  TPQ_REQUIRE(metric == TPQ_METRIC_NEG_SQ_L2 || metric == TPQ_METRIC_INNER, "max_sim_split: bad metric %d",
              metric);
  TPQ_REQUIRE(l <= 65535, "max_sim_split: batch l=%d exceeds grid.y", l);
  if (!tpq_max_sim_split_supported(d, m, n)) {
    set_error("max_sim_split: d=%d (<= 64) / slice of %lld bytes (< 2 GiB) not supported; use tpq_max_sim",
              d, (long long)d * m * 4);
    return TPQ_ERR_UNSUPPORTED;
  }
  if (m == 0) return TPQ_OK;
  const int euclid = metric == TPQ_METRIC_NEG_SQ_L2 ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d <= 16) return split::launch_split<1>(A, B, vals, inds, l, d, m, n, euclid, st);
  if (d <= 32) return split::launch_split<2>(A, B, vals, inds, l, d, m, n, euclid, st);
  if (d <= 48) return split::launch_split<3>(A, B, vals, inds, l, d, m, n, euclid, st);
  return split::launch_split<4>(A, B, vals, inds, l, d, m, n, euclid, st);
}
