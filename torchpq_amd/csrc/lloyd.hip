// One Lloyd iteration of MultiKMeans.fit on PREPARED data (tpq_lloyd_prepare / tpq_lloyd_step):
// the PQ-codebook training shape -- l sub-problems, n <= 256 centroids, d <= 64 (configs[4]).
// Replaces, per iteration, the pair  get_labels -> compute_centroids  of the reference's driver
// (torchpq/clustering/MultiKMeans.py:415-453: max_sim_tn torchpq/kernels/cuda/max_sim.cu:182-309 +
// compute_centroids torchpq/kernels/cuda/compute_centroids.cu:10-86).
//
// What "prepared" buys.  The data of a fit() never changes, only the centroids do,
// yet tpq_max_sim_select re-reads the fp32 points every iteration and spends a third of its
// instructions centring and splitting them.  tpq_lloyd_prepare does it ONCE per fit:
//   a' = s (x - mu)          mu = mean of the initial centroids (distances are translation-
//                            invariant; any fixed shift works), s = the power of two that puts
//                            max |x - mu| of the sub-problem in [2^13, 2^14)
//   a' = h + m + rho         h = fp16(a'), m = fp16(a' - h):  |rho| <= 2^-22 |a'| + eta
// and stores (h, m) in MFMA-fragment order -- tile of 32 points x k-step x piece x lane x 16 B, 4 bytes
// per element, exactly the bytes of the fp32 original -- plus |a'|^2 and |x|^2 per point.  A tile's
// B operands are then eight 16-byte loads per lane, no VALU work at all.
// fp16 pieces instead of the bf16 pieces of assign_fast.hip: two fp16 pieces carry 22 significant
// bits (two bf16: 16), so the dropped products shrink from 3 x 2^-16 to 3 x 2^-22 and the bound of the
// selection by ~4x -- a quarter of the points go to the exact re-check -- and h + m is precise
// enough (2^-22 relative: two ulps of fp32) to feed the centroid UPDATE from the same bytes.  The
// price is fp16's range: hence the per-sub-problem scale, the overflow / non-finite flags (a
// flagged sub-problem sends every point to the exact path) and eta below.
//
// Selection (same scheme as assign_fast.hip section 2b, re-derived for fp16; all in scaled-centred
// units): per 32 x 32 tile  f = sum_k (C2 a1 + C1 a2 + C1 a1) - N  on v_mfma_f32_32x32x16_f16 (C = 2 c'
// split the same way; small products first) and one bf16 MFMA for N = fl |c'|^2 (three exact bf16
// pieces against ones);  g = 2 a'.c' - |c'|^2 is what the real-number distance orders by.
//   |f - g| <= [3.03 2^-22 + (16 KS + 13) 2^-23 + (d + 1) 2^-24] (|a'| + |c'|max)^2      dropped
//              products, worst-case fp32 accumulation of all MFMA terms, the norm chain
//            + 2^-22 (|a'| + |c'|max)^2                          rounding of x - mu, c - mu
//            + eta sqrt(d) (2 |c'|max + |a'|),  eta = 2^-13     fp16 subnormals, flushed or not
//            + s^2 (d + 4) 2^-24 (|x| + |c|max)^2                the exact fp32 chain's own rounding
// delta = 1.25 x that.  A point whose two best fast values differ by more than 2 delta has its
// label decided -- the arg-max of tpq_max_sim, bit for bit; the others are listed and re-evaluated
// by the exact fp32-MFMA kernel (launch_max_sim_list, kmeans.hip) on the raw data.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "probe_fast.h"

namespace tpq {
int launch_max_sim_list(const float* A, const float* B, float* vals, int64_t* inds, int l, int d, int m, int n,
                        int euclid, const int* list, const int* count, unsigned long long* keys, float* Ac, int cap,
                        hipStream_t st);  // kmeans.hip
namespace lloyd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

#ifndef TPQ_LL_PF
#define TPQ_LL_PF 0  // update kernel: tiles ahead of an L2 prefetch by LDS-DMA (0 = none; see the kernel)
#endif
constexpr int kWaves = 8;
constexpr int kMu = 128;  // floats per sub-problem in the centring table (d <= 128)
#ifndef TPQ_LL_TILES
#define TPQ_LL_TILES 32
#endif
constexpr int kTiles = TPQ_LL_TILES;  // 32-point tiles per wave and block

static int ks_of(int d) { return (d + 15) / 16; }

// ---- prepared block --------------------------------------------------------------------------
struct PrepLayout {
  int KS;
  int64_t T;  // 32-point tiles per sub-problem
  size_t hi_off, mid_off, norms_off, mu_off, scale_off, flag_off, maxbits_off, total;
};
static PrepLayout prep_layout(int l, int d, int64_t m) {
  PrepLayout L;
  L.KS = ks_of(d);
  L.T = (m + 31) / 32;
  // hi and mid pieces in two arrays [l][T tiles][Q = ceil(KS / 2) k-step pairs][32 points][64 B]: a point's
  // 32 dimensions of a k-step pair are 64 contiguous bytes -- chunk 2 (st % 2) + half is what lane (point,
  // half) of the B operand of k-step st reads.  The coarse pass streams the hi array only; the update
  // streams one pair per wave; level 2 gathers a listed point as Q 64-byte pieces per array.  (Plain
  // MFMA-fragment order -- tile x k-step x lane x 16 B -- scatters a point over 16 cache lines: 1 KiB of
  // traffic per gathered point, level 2 at 1.6 ms instead of 0.6; plain row-major -- one 32 KS-byte row
  // per point -- makes every 64-lane load touch 32 lines: the streaming kernels turn address-unit-bound,
  // the update at 4.75 ms.  Here a 64-lane load touches 16 lines and uses half of each.)
  L.hi_off = 0;
  L.mid_off = (size_t)l * L.T * ((L.KS + 1) / 2) * 2048;
  L.norms_off = 2 * L.mid_off;
  L.mu_off = L.norms_off + (size_t)l * L.T * 32 * 8;       // [l][T * 32] float2
  L.scale_off = L.mu_off + (size_t)l * kMu * 4;            // [l][kMu] f32
  L.flag_off = L.scale_off + (size_t)l * 4;                // [l] f32
  L.maxbits_off = L.flag_off + (size_t)l * 4;              // [l] i32
  L.total = (L.maxbits_off + (size_t)l * 4 + 255) / 256 * 256;
  return L;
}

// mu[b][k] = mean over the n initial centroids of dimension k (zero beyond d)
__global__ __launch_bounds__(256) void mu_kernel(const float* __restrict__ B, float* __restrict__ mu, int d, int n) {
  __shared__ float red[256];
  const int k = blockIdx.x, b = blockIdx.y;
  const float* row = B + ((int64_t)b * d + k) * n;
  float s = 0.f;
  for (int c = threadIdx.x; c < n; c += 256) s += row[c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = red[0] / (float)n;
    mu[b * kMu + k] = (v == v && fabsf(v) <= 3.0e38f) ? v : 0.f;  // a non-finite mean: no centring (flagged below)
  }
}

// max |x - mu| per sub-problem (bits of a non-negative float: integer order == value order) and a
// flag for any non-finite element.  grid (chunks, d, l)
// `sample` > 1: only every sample-th 4-KiB run of a row is read (tpq_lloyd_prepare: the scale then leaves one bit
// of headroom and split_kernel, which sees every element, flags what exceeds it -- a full pass over 16 GB for
// a power of two was 2.8 ms of the 13.3 ms the preparation took)
__global__ __launch_bounds__(256) void maxabs_kernel(const float* __restrict__ A, const float* __restrict__ mu,
                                                    unsigned* __restrict__ maxbits, int* __restrict__ flag, int d,
                                                    int64_t m, int sample = 1) {
  const int k = blockIdx.y, b = blockIdx.z;
  const float* row = A + ((int64_t)b * d + k) * m;
  const float mk = mu[b * kMu + k];
  float mx = 0.f;
  int bad = 0;
  const int64_t per = (m + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * per, i1 = (i0 + per) < m ? (i0 + per) : m;
  if ((m & 3) == 0 && (per & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0) {
    for (int64_t i = i0 + (int64_t)threadIdx.x * 4; i < i1; i += 1024 * (int64_t)sample) {
      const float4 x = *reinterpret_cast<const float4*>(row + i);
      const float v0 = fabsf(x.x - mk), v1 = fabsf(x.y - mk), v2 = fabsf(x.z - mk), v3 = fabsf(x.w - mk);
      bad |= !(v0 <= 3.0e38f) | !(v1 <= 3.0e38f) | !(v2 <= 3.0e38f) | !(v3 <= 3.0e38f);
      mx = fmaxf(fmaxf(mx, fmaxf(v0, v1)), fmaxf(v2, v3));
    }
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256 * (int64_t)sample) {
      const float v = fabsf(row[i] - mk);
      bad |= !(v <= 3.0e38f);
      mx = fmaxf(mx, v);
    }
  }
  __shared__ float red[256];
  __shared__ int redb[256];
  red[threadIdx.x] = mx;
  redb[threadIdx.x] = bad;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + w]);
      redb[threadIdx.x] |= redb[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (red[0] > 0.f) atomicMax(maxbits + b, __float_as_uint(red[0]));
    if (redb[0]) atomicOr(flag + b, 1);
  }
}

// s[b] = 2^(13 - floor(log2 max)): max |x - mu| s in [2^13, 2^14)
// (headroom = 1: the maximum came from a sample; it lands in [2^12, 2^13) and the data may exceed it twofold)
__global__ void scale_kernel(const unsigned* __restrict__ maxbits, int* __restrict__ flag, float* __restrict__ scale,
                             int l, int headroom = 0) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= l) return;
  const float mx = __uint_as_float(maxbits[b]);
  float s = 1.f;
  if (flag[b] || !(mx <= 3.0e38f)) {
    flag[b] = 1;
  } else if (mx > 0.f) {
    int e = ilogbf(mx);
    int se = 13 - headroom - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    s = ldexpf(1.f, se);
    if (!(mx * s < 16384.f)) flag[b] = 1;  // (a clamped exponent on astronomically large data)
  }
  scale[b] = s;
}

// norms[point] = (|a'|^2, packed): the second word carries two quantities that only ever enter BOUNDS, each
// rounded UP to bf16: |x|^2 (the exact kernel's own rounding scales with it) in the high half, and
// |a' - ah|^2 -- what level 1 drops of this point -- in the low half.
__device__ __forceinline__ unsigned bf16_up(float x) {  // x >= 0 (an overflow to inf just lists the point)
  return (__float_as_uint(x) + 0xffffu) >> 16;
}
__device__ __forceinline__ float pack_bound_norms(float n2r, float n2m) {
  return __uint_as_float((bf16_up(n2r) << 16) | bf16_up(n2m));
}
__device__ __forceinline__ void unpack_bound_norms(float y, float& n2r, float& n2m) {
  const unsigned u = __float_as_uint(y);
  n2r = __uint_as_float(u & 0xffff0000u);
  n2m = __uint_as_float(u << 16);
}
constexpr int kCm = 4;  // words per sub-problem in cmax2_bits: max N, max |c|^2, max |C - Ch|^2, -

// pieces + norms.  grid (ceil(m / 256), l), 4 waves; LANE = POINT (64 consecutive points per wave = two tiles):
// every load instruction reads 256 contiguous bytes of one dimension's row, and a lane owns the 64 contiguous
// bytes of its point in each (k-step pair, piece), written as four 16-byte chunks -- a wave's stores of one pair
// are two whole 2-KiB runs.  (Round 3's kernel gave a lane (point, half of a k-step): 128-byte reads, 32-byte
// interleaved writes, 2.9 TB/s over 16 GB in + 16 GB out.)  Every element is seen here, so this is also where a
// non-finite value, or one beyond the range the (sampled) scale leaves, flags its sub-problem.
// The coarse probe's queries get the same pieces from probe_split_kernel (below), and three more things: the query as a
// ROW (the select kernel's exact step reads a query's d values; from the [d][nq] operand that is d cache lines per
// query), |x|^2 as the exact kernels sum it (fma chain over ascending k), and the query's candidate band and fp16
// scale (probe_band).
struct ProbeSplitOut {
  float* xt;                   // [m][xt_stride] fp32 row copies
  float* q2;                   // [m] |x|^2
  float* band;                 // [m]
  float* qscale;               // [m]
  const unsigned* cmax2_bits;  // the prepared centroids' maxima (kCm words)
  const int* cflag;
  float eps, eps_exact, eta;
  int xt_stride;
};
__device__ __forceinline__ void probe_band(const ProbeSplitOut& po, float s, float n2c, float n2r, float n2m, float& band,
                                           float& qscale);

__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ A, const float* __restrict__ mu,
                                                   const float* __restrict__ scale, u32x4* __restrict__ hi,
                                                   u32x4* __restrict__ mid, float2* __restrict__ norms,
                                                   int* __restrict__ flag, int d, int64_t m, int64_t T, int KS) {
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tile = i >> 5;
  if (tile >= T) return;
  const int l31 = (int)(i & 31);
  const bool iv = i < m;
  const float* Ab = A + (int64_t)b * d * m + (iv ? i : 0);
  const float* mub = mu + b * kMu;
  const float s = scale[b];
  const int Q = (KS + 1) / 2;
  float n2c = 0.f, n2r = 0.f, n2m = 0.f;
  int bad = 0;
  for (int q = 0; q < Q; ++q) {
    float x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int k = 32 * q + j;
      x[j] = (iv && k < d) ? Ab[(int64_t)k * m] : 0.f;
    }
    const int64_t fo = (((int64_t)b * T + tile) * Q + q) * 128 + l31 * 4;  // in 16-byte chunks
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f16x8 h, mm;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * q + 8 * c + j;
        const float xv = x[8 * c + j];
        const float a = (iv && k < d) ? (xv - mub[k]) * s : 0.f;
        bad |= !(fabsf(a) < 16384.f);
        const _Float16 hh = (_Float16)a;
        const float r = a - (float)hh;
        h[j] = hh;
        mm[j] = (_Float16)r;
        n2c = fmaf(a, a, n2c);
        n2r = fmaf(xv, xv, n2r);
        n2m = fmaf(r, r, n2m);
      }
      hi[fo + c] = __builtin_bit_cast(u32x4, h);
      mid[fo + c] = __builtin_bit_cast(u32x4, mm);
    }
  }
  // (a point the scale cannot hold carries an infinite norm: every bound derived from it is infinite, whoever reads it)
  norms[(int64_t)b * T * 32 + i] = make_float2(bad ? INFINITY : n2c, pack_bound_norms(n2r, n2m));
  if (__ballot(bad != 0) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag + b, 1);
}

// ---- per iteration: centroid fragments -----------------------------------------------------------
// grid (8 units, l), 64 lanes: lane (row = centroid l31 of the unit, k-group half).
// frags [l][8][2 KS + 1][64] x 16 B: fragment 0 = -N (N = fl |c'|^2) as three exact bf16 pieces at
// k = 0, 1, 2 (rows beyond n: -3e38, never first or second); fragments 1 + 2 st + q = piece q of
// C = 2 c' = 2 s (c - mu), k-step st, fp16.
__device__ __forceinline__ void split3_bf16(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}

__global__ __launch_bounds__(64) void cprep_kernel(const float* __restrict__ B, const float* __restrict__ mu,
                                                  const float* __restrict__ scale, u32x4* __restrict__ frags,
                                                  unsigned* __restrict__ cmax2_bits, int* __restrict__ cflag, int d,
                                                  int n, int KS) {
  const int unit = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  const int c = unit * 32 + l31;
  const int FPU = 2 * KS + 1;
  const float* Bb = B + (int64_t)b * d * n;
  const float s = scale[b];
  u32x4* out = frags + ((int64_t)b * gridDim.x + unit) * FPU * 64 + lane;  // (gridDim.x = 8 units per chunk of 256)
  float N = 0.f, sraw = 0.f;
  if (c < n)
    for (int k = 0; k < d; ++k) {
      const float y = Bb[(int64_t)k * n + c];
      const float cc = (y - mu[b * kMu + k]) * s;
      N = fmaf(cc, cc, N);
      sraw = fmaf(y, y, sraw);
    }
  int bad = 0;
  {
    bf16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) {
      __bf16 p1, p2, p3;
      split3_bf16(c < n ? -N : -3.0e38f, p1, p2, p3);
      f[0] = p1;
      f[1] = p2;
      f[2] = p3;
    }
    out[0] = __builtin_bit_cast(u32x4, f);
  }
  if (c < n) {
    bad |= !(N <= 3.0e38f) | !(sraw <= 3.0e38f);
    if (half == 0 && !bad) {
      atomicMax(cmax2_bits + b * kCm, __float_as_uint(N));
      atomicMax(cmax2_bits + b * kCm + 1, __float_as_uint(sraw));
    }
  }
  float c2m = 0.f;  // |C - Ch|^2: what level 1 drops of this centroid
  for (int st = 0; st < KS; ++st) {
    f16x8 h, mm;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * st + 8 * half + j;
      const float C = (k < d && c < n) ? 2.f * ((Bb[(int64_t)k * n + c] - mu[b * kMu + k]) * s) : 0.f;
      bad |= !(fabsf(C) <= 65000.f);  // beyond fp16's range (or NaN): the whole sub-problem goes exact
      const _Float16 hh = (_Float16)C;
      const float r = C - (float)hh;
      h[j] = hh;
      mm[j] = (_Float16)r;
      c2m = fmaf(r, r, c2m);
    }
    out[(1 + 2 * st) * 64] = __builtin_bit_cast(u32x4, h);
    out[(2 + 2 * st) * 64] = __builtin_bit_cast(u32x4, mm);
  }
  c2m += __shfl_xor(c2m, 32, 64);
  if (half == 0 && c < n && !bad) atomicMax(cmax2_bits + b * kCm + 2, __float_as_uint(c2m));
  if (bad) atomicOr(cflag + b, 1);
}

// ---- top-2 of fast values (assign_fast.hip) -----------------------------------------------------
template <int CL>
__device__ __forceinline__ void take_top2(float& b1, float& b2, int& bi, float v) {
  static_assert(CL >= 0 && CL <= 64, "inline constant");
  asm volatile(
      "v_cmp_ngt_f32 vcc, %3, %0\n\t"
      "v_cndmask_b32 %2, %4, %2, vcc\n\t"
      "v_med3_f32 %1, %0, %1, %3\n\t"
      "v_max_f32 %0, %3, %0"
      : "+v"(b1), "+v"(b2), "+v"(bi)
      : "v"(v), "n"(CL)
      : "vcc");
}
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

// Key epilogue: the accumulator register number r (0..15: which of the lane's 16 centroid rows of
// the unit) replaces the value's low 4 mantissa bits, so the running best carries its own index
// and no compare / select is needed: per PAIR of values  t = med3(b1, k0, k1); b1 = max3(b1, k0, k1);
// b2 = max(b2, t)  (the second best of {b1 >= b2, k0, k1} is max(med3(b1, k0, k1), b2)).  5 VALU per
// two values against 8; the 2^-19 |v| the keys are off by is part of the bound (StepArgs::eps).
template <int R0>
__device__ __forceinline__ void take_keys_quad(float& p1, float& p2, float& q1, float& q2, float v0, float v1,
                                               float v2, float v3) {
  static_assert(R0 >= 0 && R0 + 3 <= 15, "inline constants");
  float k0, k1, k2, k3, t0, t1;
  asm volatile(
      "v_and_or_b32 %4, %10, -16, %14\n\t"
      "v_and_or_b32 %5, %11, -16, %15\n\t"
      "v_and_or_b32 %6, %12, -16, %16\n\t"
      "v_and_or_b32 %7, %13, -16, %17\n\t"
      "v_med3_f32 %8, %0, %4, %5\n\t"
      "v_med3_f32 %9, %2, %6, %7\n\t"
      "v_max3_f32 %0, %0, %4, %5\n\t"
      "v_max3_f32 %2, %2, %6, %7\n\t"
      "v_max_f32 %1, %1, %8\n\t"
      "v_max_f32 %3, %3, %9"
      : "+v"(p1), "+v"(p2), "+v"(q1), "+v"(q2), "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3), "=&v"(t0), "=&v"(t1)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(R0), "n"(R0 + 1), "n"(R0 + 2), "n"(R0 + 3));
}

template <int R0>
__device__ __forceinline__ void take_keys_pair(float& p1, float& p2, float v0, float v1) {
  static_assert(R0 >= 0 && R0 + 1 <= 15, "inline constants");
  float k0, k1, t0;
  asm volatile(
      "v_and_or_b32 %2, %5, -16, %7\n\t"
      "v_and_or_b32 %3, %6, -16, %8\n\t"
      "v_med3_f32 %4, %0, %2, %3\n\t"
      "v_max3_f32 %0, %0, %2, %3\n\t"
      "v_max_f32 %1, %1, %4"
      : "+v"(p1), "+v"(p2), "=&v"(k0), "=&v"(k1), "=&v"(t0)
      : "v"(v0), "v"(v1), "n"(R0), "n"(R0 + 1));
}

// Level 1's keys carry 6 bits -- register number + 16 x (unit mod 4) -- so that the unit of the best value
// needs no bookkeeping of its own (which half of the tile's units it came from is one compare per tile).
// 2^-17 |v| off: in level 1's bound.
// The tagging itself is plain C++ (the compiler selects v_and_or_b32): these instructions READ MFMA
// results right behind the MFMAs, and the wait states that takes are only inserted for instructions
// the hazard recogniser can see -- as operands of an asm block the accumulators were read too early
// (labels wrong, differently on every run).
template <int TAG>
__device__ __forceinline__ float key6(float v) {
  return __int_as_float((int)((__float_as_uint(v) & 0xffffffc0u) | (unsigned)TAG));
}
__device__ __forceinline__ void top2_keys_pair(float& p1, float& p2, float k0, float k1) {
  float t0;
  asm volatile(
      "v_med3_f32 %2, %0, %3, %4\n\t"
      "v_max3_f32 %0, %0, %3, %4\n\t"
      "v_max_f32 %1, %1, %2"
      : "+v"(p1), "+v"(p2), "=&v"(t0)
      : "v"(k0), "v"(k1));
}

// ---- the cascade on prepared pieces ------------------------------------------------------------------
// Level 1 (coarse_kernel): ONE product per k-step -- f0 = sum_k Ch ah - N on the hi pieces only (half the
// bytes, 5 MFMAs per 32 x 32 tile instead of 13).  |f0 - g| carries the dropped pieces,
//   (2^-11 + 2^-23) (|a'| + |c'|max)^2      (|a - ah| <= 2^-11 |a|, |C - Ch| <= 2^-11 |C|, 2 |c'||a'| <= (.)^2 / 2)
// in place of 3.03 2^-22 (.)^2: the bound is ~36x wider and 5-13 % of the points stay undecided.
// Level 2 (refine_kernel): those points, gathered through the level-1 list, with all three products
// (the bound of the header comment): 0.2-0.7 % stay undecided.
// Level 3: the exact fp32 kernel over the level-2 list (launch_max_sim_list, kmeans.hip).
// Every level decides a point only when its two best fast values are further apart than twice its
// own rigorous bound, so the labels are tpq_max_sim's whatever the split between the levels.
struct StepArgs {
  const u32x4* hi;             // [l][T][Q][32 points][64 B]
  const u32x4* mid;            // likewise
  const float2* norms;         // [l][T * 32]: (|a'|^2, |x|^2)
  const u32x4* frags;          // [l][8][2 KS + 1][64]
  const unsigned* cmax2_bits;  // [l][kCm]: max N, max |c|^2, max |C - Ch|^2
  const float* scale;          // [l]
  const int* flag;             // [l] data not finite / out of range (prepare)
  const int* cflag;            // [l] centroids out of fp16 range (this iteration)
  int64_t* inds;               // [l][m]
  float* vals;                 // optional [l][m]
  const int* list_in;          // level 2: [l][m] points to refine, count_in [l]
  const int* count_in;
  int* list;                   // [l][m] points this level leaves undecided
  int* count;                  // [l]
  int m;
  int64_t T;
  float eps, eps_exact, eta;   // eps: this level's fast-path bound, relative to (|a'| + |c'|max)^2; eta times sqrt(d)
  int level;                   // 1: the dropped pieces are bounded per point (emit), on top of eps
  float* thr;                  // chunked level 1, candidate route: [thr_cap] threshold of the listed point (or null)
  int thr_cap;
  // more than 256 centroids (tpq_coarse_assign): blockIdx.y = CHUNK of 256 centroids (all chunks in one
  // launch: one chunk's blocks alone fill half the chip); a chunk's (best, second) and in-chunk index of
  // every point go to part_*[chunk][point or list position], decide_kernel folds the chunks and decides
  float2* part_b;              // [chunks][m]  (nullptr: a single chunk, decided in the kernel)
  uint8_t* part_i;             // [chunks][m]
  int chunk_frag_stride;       // 16-byte units between the fragment blocks of consecutive chunks
};

// label, value and -- unless the two best fast values are more than 2 delta apart -- a list entry.
// Called by all lanes of the wave.  The list is staged in LDS (one LDS atomic per wave) and flushed
// once per block (flush_list): a RETURNING global atomic per tile put a memory round trip -- and,
// vmcnt being in order, the wait for every load issued before it -- into each tile of level 1,
// where 92 % of the tiles hold an undecided point (19 ms instead of 2).
template <int CAP>
struct BlockListT {
  int n;
  int base;
  int item[CAP];
};
template <int CAP>
__device__ __forceinline__ void emit(const StepArgs& a, BlockListT<CAP>* bl, int b, int lane, bool valid, int64_t fi,
                                     int idx, float B1, float B2, float2 n2, float s, float cn, float cnr,
                                     float inv_s2, bool exact_all, float c2, int64_t part_slot = -1) {
  if (a.part_b != nullptr) {  // chunked: this chunk's result of the point; decide_kernel does the rest
    if (valid) {
      a.part_b[part_slot] = make_float2(B1, B2);
      a.part_i[part_slot] = (uint8_t)idx;
    }
    return;
  }
  // (v_sqrt_f32: 1 ulp; the norms only scale the bound, whose 1.25 covers it)
  float n2r, n2m;
  unpack_bound_norms(n2.y, n2r, n2m);
  const float an = __builtin_amdgcn_sqrtf(n2.x), anr = __builtin_amdgcn_sqrtf(n2r) * s;
  const float t1 = an + cn, t2 = anr + cnr * s;
  // Level 1 drops the products with the mid pieces: |sum (a C - ah Ch)| <= |a' - ah| (|Ch|max + |C - Ch|max) +
  // |a'| |C - Ch|max with what was ACTUALLY dropped of this point and of the worst centroid (|Ch| <= (1 + 2^-11)
  // 2 |c'|) -- about 0.4 of the worst case 2^-11 (|a'| + |c'|max)^2, and 7.5 % undecided points become 3 %.
  // (c2 = max |C - Ch|, read ONCE by the caller: a load here, in the tile loop, sits behind the prefetched pieces
  // on the in-order vmcnt and cost level 1 18 % -- or < 0 at level 2, which drops nothing)
  float dropped = 0.f;
  if (c2 >= 0.f) {
    const float a2 = __builtin_amdgcn_sqrtf(n2m);
    dropped = a2 * (2.002f * cn + c2) + 1.001f * an * c2;
  }
  float delta = 1.25f * (dropped + a.eps * t1 * t1 + a.eta * (2.f * cn + an) + a.eps_exact * t2 * t2);
  if (exact_all) delta = INFINITY;
  if (valid) {
    a.inds[(int64_t)b * a.m + fi] = idx;
    if (a.vals) a.vals[(int64_t)b * a.m + fi] = (B1 - n2.x) * inv_s2;
  }
  const bool listed = valid && !(B1 - B2 > 2.f * delta);
  const unsigned long long mk = __ballot(listed);
  if (mk) {
    const int leader = __ffsll((long long)mk) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&bl->n, __popcll(mk));  // LDS
    base = __shfl(base, leader, 64);
    if (listed) bl->item[base + __popcll(mk & ((1ull << lane) - 1ull))] = (int)fi;
  }
}
// end of the block: reserve [base, base + n) of the sub-problem's list with one global atomic, copy
template <int CAP>
__device__ __forceinline__ void flush_list(const StepArgs& a, BlockListT<CAP>* bl, int b) {
  __syncthreads();
  if (threadIdx.x == 0) bl->base = bl->n ? atomicAdd(a.count + b, bl->n) : 0;
  __syncthreads();
  const int n = bl->n, base = bl->base;
  for (int i = threadIdx.x; i < n; i += kWaves * 64) a.list[(int64_t)b * a.m + base + i] = bl->item[i];
}

// ---- level 1 -----------------------------------------------------------------------------------------
// A wave owns WIDE tiles of 64 points (two MFMA column tiles sharing every A operand: at one A
// operand per MFMA the centroid fragments alone would take the whole LDS bandwidth -- 1 KiB per
// 32-cycle MFMA per SIMD = 128 B/clk/CU); the two accumulators of a k-step are independent, so no
// MFMA waits for the one before it.  LDS holds -N and the hi pieces of the centroids only (40 KiB).
constexpr int kWide = kTiles / 2;  // wide tiles per wave and block

constexpr int kCoarseList = kWaves * kWide * 64;  // points a level-1 block decides = capacity of its staged list

template <int KS>
__global__ __launch_bounds__(kWaves * 64, 2) void coarse_kernel(StepArgs a) {
  constexpr int FPU = 2 * KS + 1;  // fragments per unit in global memory
  constexpr int FL = KS + 1;       // ... in LDS
  constexpr int Q = (KS + 1) / 2;
  typedef BlockListT<kCoarseList> BL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool chunked = a.part_b != nullptr;
  const int b = chunked ? 0 : blockIdx.y, chunk = chunked ? blockIdx.y : 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int m = a.m;
  BL* bl = reinterpret_cast<BL*>(smem + 8 * FL * 1024);
  if (threadIdx.x == 0) bl->n = 0;
  {
    const char* src = reinterpret_cast<const char*>(a.frags) + (size_t)b * 8 * FPU * 1024 +
                      (size_t)chunk * a.chunk_frag_stride * 16;
    for (int f = wave; f < 8 * FL; f += kWaves) {
      const int unit = f / FL, j = f % FL;
      const int sf = unit * FPU + (j ? 2 * j - 1 : 0);  // -N, then the hi piece of k-step j - 1
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + sf * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + f * 1024), 16, 0, 0);
    }
  }
  const int64_t slice = a.T * Q * 2048;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.hi) + (size_t)b * slice), 0, (int)slice, 0x00020000);
  const float2* __restrict__ nrm = a.norms + (int64_t)b * a.T * 32;
  auto wide_of = [&](int t) -> int64_t { return ((int64_t)blockIdx.x * kWide + t) * kWaves + wave; };
  auto frag_voff = [&](int t) -> int {
    const int64_t wt = wide_of(t);
    return (t < kWide && 2 * wt < a.T) ? (int)(2 * wt * Q * 2048) + l31 * 64 + half * 16 : 0x7ffffff0;
  };
  f16x8 xsb[2][2][KS];  // [buffer][column tile][k-step]
  float2 n2b[2][2];     // [buffer][column tile]
  // fragment e of the wide tile: column tile e / KS (the next tile), k-step e % KS
  auto load_frag = [&](int voff, auto e_c, f16x8 (&dst)[2][KS]) {
    constexpr int e = decltype(e_c)::value, ct = e / KS, st = e % KS;
    dst[ct][st] = __builtin_bit_cast(
        f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ct * Q * 2048 + (st >> 1) * 2048 + (st & 1) * 32, 0));
  };
  auto load_norm = [&](int t, int ct) -> float2 {
    const int64_t tile = 2 * wide_of(t) + ct;  // (clamped: a tile beyond the range reads tile 0's norms; never used)
    return nrm[((t < kWide && tile < a.T) ? tile : 0) * 32 + l31];
  };
  {
    const int voff = frag_voff(0);
    static_for<0, 2 * KS>([&](auto e_c) { load_frag(voff, e_c, xsb[0]); });
    n2b[0][0] = load_norm(0, 0);
    n2b[0][1] = load_norm(0, 1);
  }
  __syncthreads();  // fragments (vmcnt(0) of the DMA) are in LDS
  const u32x4* fp = reinterpret_cast<const u32x4*>(smem) + lane;
  auto ldsf = [&](const u32x4* p) -> f16x8 { return __builtin_bit_cast(f16x8, *p); };

  // ONE accumulator per column tile.  A unit = its 2 (KS + 1) MFMAs -- the two tiles in turn on every A
  // operand, so no MFMA waits for the one before it and the centroid fragments cross the LDS port once
  // per TWO MFMAs -- then the top-2 update of its 2 x 16 values; the SIMD's other wave has its MFMAs
  // meanwhile.  The A operands run through a three-slot ring two k-steps ahead (all KS of a unit in
  // registers: 32 of them at d = 128).  Tried on the way (C5, all within 3 % of each other: the kernel is
  // bound by the VALU work of the update, not by its schedule): the two tiles half a unit out of phase
  // (updates of one between the MFMAs of the other); four waves per SIMD without register prefetch;
  // two accumulator SETS (updating unit U - 1 between the MFMAs of unit U): 256 VGPRs + 53 spilled
  // around the per-tile epilogue -- and a scratch reload waits, vmcnt being in order, for the piece
  // loads issued before it.
  f32x16 acc[2];
  float b1[2] = {-INFINITY, -INFINITY}, b2[2] = {-INFINITY, -INFINITY};
  float b1h[2] = {-INFINITY, -INFINITY};  // the best after units 0..3
  // A operands: k-steps 0 and 1 of a unit in a0 / a1 -- re-loaded for the NEXT unit as soon as this unit's
  // MFMAs have taken them --, k-steps >= 2 through a three-slot ring two k-steps ahead
  f16x8 a0 = ldsf(fp + 1 * 64), a1 = a0, aring[3];
  if constexpr (KS > 1) a1 = ldsf(fp + 2 * 64);
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }
  const float s = a.scale[b];
  const float cn = sqrtf(__uint_as_float(a.cmax2_bits[b * kCm])), cnr = sqrtf(__uint_as_float(a.cmax2_bits[b * kCm + 1]));
  const float c2n = a.level == 1 ? sqrtf(__uint_as_float(a.cmax2_bits[b * kCm + 2])) : -1.f;
  const bool exact_all = (a.flag[b] | a.cflag[b]) != 0;
  const float inv_s2 = (1.f / s) * (1.f / s);

  auto finish = [&](int ct, int64_t tile, float2 n2) {
    const int tag = __float_as_int(b1[ct]) & 63, r0 = tag & 15;
    // a best key found in units 4..7 is greater than the best of units 0..3 (equal keys: b2 == b1, listed)
    const int unit = (tag >> 4) + (b1[ct] > b1h[ct] ? 4 : 0);
    int idx = unit * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * half;
    const float m1 = b1[ct], m2 = b2[ct];
    const float o1 = __shfl_xor(m1, 32, 64), o2 = __shfl_xor(m2, 32, 64);
    const int oi = __shfl_xor(idx, 32, 64);
    const float B1 = fmaxf(m1, o1);
    const float B2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2));
    if (o1 > m1 || (o1 == m1 && oi < idx)) idx = oi;
    const int64_t fi = tile * 32 + l31;
    emit(a, bl, b, lane, half == 0 && fi < m, fi, idx, B1, B2, n2, s, cn, cnr, inv_s2, exact_all, c2n,
         (int64_t)chunk * m + fi);
  };
  using std::integral_constant;

  auto unit = [&](auto u_c, int voff_next, const f16x8 (&xs)[2][KS], f16x8 (&xsn)[2][KS]) {
    constexpr int U = decltype(u_c)::value;
    const u32x4* up = fp + U * FL * 64;
    const u32x4* upn = fp + ((U + 1) & 7) * FL * 64;  // the next unit (unit 0 of the next tile after 7)
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16x8 cfrag = __builtin_bit_cast(bf16x8, up[0]);
    if constexpr (U < 4) {  // the next wide tile's hi pieces: 2 KS 16-byte loads over units 0..3
      constexpr int l0 = (U * 2 * KS) / 4, l1 = ((U + 1) * 2 * KS) / 4;
      static_for<l0, l1>([&](auto e_c) { load_frag(voff_next, e_c, xsn); });
    }
    if constexpr (U == 4) {
      b1h[0] = b1[0];
      b1h[1] = b1[1];
    }
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      if constexpr (st + 2 < KS) aring[(st + 2) % 3] = ldsf(up + (1 + st + 2) * 64);
      if constexpr (st == 0) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xs[0][0], zero, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xs[1][0], zero, 0, 0, 0);
        a0 = ldsf(upn + 1 * 64);
      } else if constexpr (st == 1) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xs[0][1], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xs[1][1], acc[1], 0, 0, 0);
        a1 = ldsf(upn + 2 * 64);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[st % 3], xs[0][st], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[st % 3], xs[1][st], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, acc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 16>([&](auto q_c) {  // 16 register pairs, the two column tiles in turn
      constexpr int q = decltype(q_c)::value, ct = q & 1, pq = q >> 1;
      top2_keys_pair(b1[ct], b2[ct], key6<2 * pq + 16 * (U & 3)>(acc[ct][2 * pq]),
                     key6<2 * pq + 1 + 16 * (U & 3)>(acc[ct][2 * pq + 1]));
    });
    __builtin_amdgcn_sched_barrier(0);
  };

  auto tile = [&](int t, auto cb_c) {
    constexpr int CB = decltype(cb_c)::value, NX = 1 - CB;
    const int voff_next = frag_voff(t + 1);
    n2b[NX][0] = load_norm(t + 1, 0);
    n2b[NX][1] = load_norm(t + 1, 1);
    b1[0] = b1[1] = b2[0] = b2[1] = -INFINITY;
    static_for<0, 8>([&](auto u_c) { unit(u_c, voff_next, xsb[CB], xsb[NX]); });
    const int64_t wt = wide_of(t);
    finish(0, 2 * wt, n2b[CB][0]);
    finish(1, 2 * wt + 1, n2b[CB][1]);
  };
#pragma unroll 1
  for (int t = 0; t < kWide; t += 2) {
    if (2 * ((int64_t)blockIdx.x * kWide + t) * kWaves >= a.T) break;
    tile(t, integral_constant<int, 0>{});
    if (t + 1 >= kWide || 2 * ((int64_t)blockIdx.x * kWide + t + 1) * kWaves >= a.T) break;
    tile(t + 1, integral_constant<int, 1>{});
  }
  if (!chunked) flush_list(a, bl, b);
}

// ---- level 2 -----------------------------------------------------------------------------------------
// The three-product selection (assign_fast.hip section 2b's loop order, fp16 pieces) over the points of
// the level-1 list: a tile is 32 LISTED points, their pieces gathered from the hi and mid arrays (64
// contiguous bytes per point, k-step pair and array).  The grid covers the worst case (every point listed); blocks beyond the
// list leave at once.
constexpr int kTilesR = 8;  // 32-point tiles per wave and block: 2048 listed points per block (many small blocks:
                            // the list is a few percent of the points and its length is only known on the device)
constexpr int kRefineList = kTilesR * kWaves * 32;  // points a level-2 block decides
template <int KS>
__global__ __launch_bounds__(kWaves * 64, 2) void refine_kernel(StepArgs a) {
  constexpr int FPU = 2 * KS + 1;
  constexpr int NM = 3 * KS + 1;  // MFMAs per unit
  constexpr bool PF = KS <= 4;    // the next tile's pieces prefetched into a second register set (d <= 64);
                                  // beyond, that set does not fit: the pieces are loaded when the tile is done
  typedef BlockListT<kRefineList> BL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool chunked = a.part_b != nullptr;
  const int b = chunked ? 0 : blockIdx.y, chunk = chunked ? blockIdx.y : 0;
  const int m = a.m;
  int cnt = a.count_in[b];
  cnt = cnt < m ? cnt : m;
  if ((int64_t)blockIdx.x * kTilesR * kWaves * 32 >= cnt) return;  // block-uniform
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  BL* bl = reinterpret_cast<BL*>(smem + 8 * FPU * 1024);
  if (threadIdx.x == 0) bl->n = 0;
  {
    const char* src = reinterpret_cast<const char*>(a.frags) + (size_t)b * 8 * FPU * 1024 +
                      (size_t)chunk * a.chunk_frag_stride * 16;
    for (int f = wave; f < 8 * FPU; f += kWaves)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + f * 1024), 16, 0, 0);
  }
  constexpr int Q = (KS + 1) / 2;
  const int64_t slice = a.T * Q * 2048;
  const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.hi) + (size_t)b * slice), 0, (int)slice, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mid = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.mid) + (size_t)b * slice), 0, (int)slice, 0x00020000);
  const float2* __restrict__ nrm = a.norms + (int64_t)b * a.T * 32;
  const int* __restrict__ lst = a.list_in + (int64_t)b * m;
  // tile t of this wave = positions [32 tile, 32 tile + 32) of the list
  auto pos_of = [&](int t) -> int64_t { return (((int64_t)blockIdx.x * kTilesR + t) * kWaves + wave) * 32 + l31; };
  auto point_of = [&](int t) -> int {
    const int64_t pos = pos_of(t);
    return (t < kTilesR && pos < cnt) ? lst[pos] : -1;
  };
  auto voff_of = [&](int p) -> int {
    return p >= 0 ? (p >> 5) * (Q * 2048) + (p & 31) * 64 + half * 16 : 0x7ffffff0;
  };
  f16x8 xs[KS][2], xsn[PF ? KS : 1][2];
  auto load_frag = [&](int voff, auto e_c, f16x8 (&dst)[KS][2]) {
    constexpr int e = decltype(e_c)::value, st = e >> 1;
    dst[st][e & 1] = __builtin_bit_cast(
        f16x8, __builtin_amdgcn_raw_buffer_load_b128((e & 1) ? rs_mid : rs_hi, voff, (st >> 1) * 2048 + (st & 1) * 32, 0));
  };
  auto load_norm = [&](int p) -> float2 { return nrm[p >= 0 ? p : 0]; };  // (clamped, never used when p < 0)
  int p_cur = point_of(0), p_nxt = point_of(1), p_nx2 = -1, p_prev = -1;
  float2 n2cur = load_norm(p_cur), n2nxt = make_float2(0.f, 0.f), n2prev = make_float2(0.f, 0.f);
  {
    const int voff = voff_of(p_cur);
    static_for<0, 2 * KS>([&](auto e_c) { load_frag(voff, e_c, xs); });
  }
  __syncthreads();  // fragments (vmcnt(0) of the DMA) are in LDS
  const u32x4* fp = reinterpret_cast<const u32x4*>(smem) + lane;
  auto ldsf = [&](const u32x4* p) -> f16x8 { return __builtin_bit_cast(f16x8, *p); };

  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = -3.0e38f;
  float b1[2] = {-INFINITY, -INFINITY}, b2[2] = {-INFINITY, -INFINITY};
  int bu[2] = {0, 0};
  f16x8 c1k[KS], c2r[3];
  c1k[0] = ldsf(fp + 1 * 64);
  c2r[0] = ldsf(fp + 2 * 64);
  if constexpr (KS > 1) {
    c1k[1] = ldsf(fp + 3 * 64);
    c2r[1] = ldsf(fp + 4 * 64);
  }
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }
  const float s = a.scale[b];
  const float cn = sqrtf(__uint_as_float(a.cmax2_bits[b * kCm])), cnr = sqrtf(__uint_as_float(a.cmax2_bits[b * kCm + 1]));
  const float c2n = a.level == 1 ? sqrtf(__uint_as_float(a.cmax2_bits[b * kCm + 2])) : -1.f;
  const bool exact_all = (a.flag[b] | a.cflag[b]) != 0;
  const float inv_s2 = (1.f / s) * (1.f / s);

  auto finish_tile = [&](int p, float2 n2, int64_t pos) {
    const int r0 = __float_as_int(b1[0]) & 15, r1 = __float_as_int(b1[1]) & 15;
    const int ia = bu[0] * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * half;
    const int ib = bu[1] * 32 + (r1 & 3) + 8 * (r1 >> 2) + 4 * half;
    const bool tb = b1[1] > b1[0] || (b1[1] == b1[0] && ib < ia);
    int idx = tb ? ib : ia;
    const float m1 = fmaxf(b1[0], b1[1]);
    const float m2 = fmaxf(fminf(b1[0], b1[1]), fmaxf(b2[0], b2[1]));
    const float o1 = __shfl_xor(m1, 32, 64), o2 = __shfl_xor(m2, 32, 64);
    const int oi = __shfl_xor(idx, 32, 64);
    const float B1 = fmaxf(m1, o1);
    const float B2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2));
    if (o1 > m1 || (o1 == m1 && oi < idx)) idx = oi;
    emit(a, bl, b, lane, half == 0 && p >= 0, p, idx, B1, B2, n2, s, cn, cnr, inv_s2, exact_all, c2n,
         (int64_t)chunk * m + pos);
  };

  auto unit = [&](auto u_c, f32x16& acc, const f32x16& fin, int voff_next, const f16x8 (&xs)[KS][2],
                  f16x8 (&xsn)[PF ? KS : 1][2]) {
    constexpr int U = decltype(u_c)::value, FU = (U + 7) & 7;
    const u32x4* up = fp + U * FPU * 64;
    const u32x4* upn = fp + ((U + 1) & 7) * FPU * 64;  // the next unit (unit 0 of the next tile after 7)
    const float before0 = b1[0], before1 = b1[1];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16x8 cfrag = __builtin_bit_cast(bf16x8, up[0]);
    auto fill = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      if constexpr (mi >= 2) {  // 8 register pairs of the previous unit's values over gaps 2 .. NM - 1
        constexpr int lo = ((mi - 2) * 16) / (NM - 2), hi = ((mi - 1) * 16) / (NM - 2);
        static_for<0, 8>([&](auto q_c) {
          constexpr int q = decltype(q_c)::value;
          if constexpr (2 * q + 1 >= lo && 2 * q + 1 < hi)
            take_keys_pair<2 * q>(b1[q & 1], b2[q & 1], fin[2 * q], fin[2 * q + 1]);
        });
      }
      if constexpr (U < 4 && mi == 0 && PF) {  // the next tile's pieces: 2 KS gathered 16-byte loads over units 0..3
        constexpr int l0 = (U * 2 * KS) / 4, l1 = ((U + 1) * 2 * KS) / 4;
        static_for<l0, l1>([&](auto e_c) {
          constexpr int e = decltype(e_c)::value, st = e >> 1;
          xsn[st][e & 1] = __builtin_bit_cast(
              f16x8, __builtin_amdgcn_raw_buffer_load_b128((e & 1) ? rs_mid : rs_hi, voff_next,
                                                           (st >> 1) * 2048 + (st & 1) * 32, 0));
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // small products first: corrections (C2 a1, C1 a2), main (C1 a1), then -N
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      if constexpr (st + 2 < KS) {
        c1k[st + 2] = ldsf(up + (1 + (st + 2) * 2) * 64);
        c2r[(st + 2) % 3] = ldsf(up + (2 + (st + 2) * 2) * 64);
      }
      if constexpr (st == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c2r[0], xs[0][0], zero, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c2r[st % 3], xs[st][0], acc, 0, 0, 0);
      }
      fill(std::integral_constant<int, 2 * st>{});
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1k[st], xs[st][1], acc, 0, 0, 0);
      fill(std::integral_constant<int, 2 * st + 1>{});
    });
    c2r[0] = ldsf(upn + 2 * 64);
    if constexpr (KS > 1) c2r[1] = ldsf(upn + (2 + 2) * 64);
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1k[st], xs[st][0], acc, 0, 0, 0);
      if constexpr (st < 2) c1k[st] = ldsf(upn + (1 + st * 2) * 64);
      fill(std::integral_constant<int, 2 * KS + st>{});
    });
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, acc, 0, 0, 0);
    fill(std::integral_constant<int, 3 * KS>{});
    bu[0] = b1[0] > before0 ? FU : bu[0];
    bu[1] = b1[1] > before1 ? FU : bu[1];
  };
  using std::integral_constant;

  bool have_prev = false;
  int t_last = 0;
  auto tile = [&](int t, f16x8 (&cur)[KS][2], f16x8 (&nxt)[PF ? KS : 1][2]) {
    const int voff_next = voff_of(p_nxt);
    n2nxt = load_norm(p_nxt);
    p_nx2 = point_of(t + 2);
    unit(integral_constant<int, 0>{}, accA, accB, voff_next, cur, nxt);
    if (have_prev) finish_tile(p_prev, n2prev, pos_of(t - 1));
    b1[0] = b1[1] = b2[0] = b2[1] = -INFINITY;
    bu[0] = bu[1] = 0;
    unit(integral_constant<int, 1>{}, accB, accA, voff_next, cur, nxt);
    unit(integral_constant<int, 2>{}, accA, accB, voff_next, cur, nxt);
    unit(integral_constant<int, 3>{}, accB, accA, voff_next, cur, nxt);
    unit(integral_constant<int, 4>{}, accA, accB, voff_next, cur, nxt);
    unit(integral_constant<int, 5>{}, accB, accA, voff_next, cur, nxt);
    unit(integral_constant<int, 6>{}, accA, accB, voff_next, cur, nxt);
    unit(integral_constant<int, 7>{}, accB, accA, voff_next, cur, nxt);
    if constexpr (!PF) static_for<0, 2 * KS>([&](auto e_c) { load_frag(voff_next, e_c, cur); });
    p_prev = p_cur;
    p_cur = p_nxt;
    p_nxt = p_nx2;
    n2prev = n2cur;
    n2cur = n2nxt;
    have_prev = true;
    t_last = t;
  };
#pragma unroll 1
  for (int t = 0; t < kTilesR; t += 2) {
    if (((int64_t)blockIdx.x * kTilesR + t) * kWaves * 32 >= cnt) break;
    tile(t, xs, xsn);
    if (t + 1 >= kTilesR || ((int64_t)blockIdx.x * kTilesR + t + 1) * kWaves * 32 >= cnt) break;
    if constexpr (PF) {
      tile(t + 1, xsn, xs);
    } else {
      tile(t + 1, xs, xsn);
    }
  }
  if (have_prev) {  // the last unit of the last tile
    const float before0 = b1[0], before1 = b1[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r >> 1) & 1;
      const float v = __int_as_float((__float_as_int(accB[r]) & ~15) | r);
      const float t = fminf(v, b1[c]);
      b1[c] = fmaxf(v, b1[c]);
      b2[c] = fmaxf(b2[c], t);
    }
    bu[0] = b1[0] > before0 ? 7 : bu[0];
    bu[1] = b1[1] > before1 ? 7 : bu[1];
    finish_tile(p_prev, n2prev, pos_of(t_last));
  }
  if (!chunked) flush_list(a, bl, b);
}

// ---- level 2, many centroids --------------------------------------------------------------------------
// The loop order of assign_fast.hip: a wave keeps ITS 32 listed points (hi and mid pieces, gathered once)
// in registers for the whole sweep and ALL centroid chunks stream through a double-buffered LDS ring
// (half a chunk = 4 units = 128 centroids per buffer, LDS-DMA, one barrier per half chunk); the running
// top-2 never leaves the registers.  (refine_kernel per chunk re-gathers the points for every chunk and
// leaves each wave waiting for its gathers: 2.25 ms for 8 % of 1 M points x 16 384 centroids.)
constexpr int kStreamList = kWaves * 32;

template <int KS>
__global__ __launch_bounds__(kWaves * 64, 2) void refine_stream_kernel(StepArgs a, int n_half) {
  constexpr int FPU = 2 * KS + 1;
  constexpr int NM = 3 * KS + 1;  // MFMAs per unit
  constexpr int HB = 4 * FPU * 1024;  // bytes of a half chunk of fragments
  constexpr int Q = (KS + 1) / 2;
  typedef BlockListT<kStreamList> BL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int m = a.m;
  int cnt = a.count_in[0];
  cnt = cnt < m ? cnt : m;
  if ((int64_t)blockIdx.x * kWaves * 32 >= cnt) return;  // block-uniform
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  BL* bl = reinterpret_cast<BL*>(smem + 2 * HB);
  if (threadIdx.x == 0) bl->n = 0;
  auto stage = [&](int h) {  // half chunk h -> buffer h & 1 (the fragment blocks of the chunks are contiguous)
    const char* src = reinterpret_cast<const char*>(a.frags) + (size_t)h * HB;
    char* dst = smem + (h & 1) * HB;
    for (int f = wave; f < 4 * FPU; f += kWaves)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
  };
  stage(0);
  const int64_t slice = a.T * Q * 2048;
  const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.hi)), 0, (int)slice, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mid = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.mid)), 0, (int)slice, 0x00020000);
  const int64_t pos = ((int64_t)blockIdx.x * kWaves + wave) * 32 + l31;
  const int p = pos < cnt ? a.list_in[pos] : -1;
  const float2 n2 = a.norms[p >= 0 ? p : 0];
  f16x8 xs[KS][2];
  {
    const int voff = p >= 0 ? (p >> 5) * (Q * 2048) + (p & 31) * 64 + half * 16 : 0x7ffffff0;
    static_for<0, 2 * KS>([&](auto e_c) {
      constexpr int e = decltype(e_c)::value, st = e >> 1;
      xs[st][e & 1] = __builtin_bit_cast(
          f16x8, __builtin_amdgcn_raw_buffer_load_b128((e & 1) ? rs_mid : rs_hi, voff, (st >> 1) * 2048 + (st & 1) * 32, 0));
    });
  }
  auto ldsf = [&](const u32x4* q) -> f16x8 { return __builtin_bit_cast(f16x8, *q); };
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = -3.0e38f;
  float b1[2] = {-INFINITY, -INFINITY}, b2[2] = {-INFINITY, -INFINITY};
  int bu[2] = {0, 0};
  f16x8 c1k[KS], c2r[3];
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // unit U of the half chunk in `base`; the values of the unit before it (`fin`, global unit number gprev)
  // go through the top-2 update between the MFMAs
  auto unit = [&](auto u_c, const u32x4* base, f32x16& acc, const f32x16& fin, int gprev) {
    constexpr int U = decltype(u_c)::value;
    const u32x4* up = base + U * FPU * 64;
    const float before0 = b1[0], before1 = b1[1];
    const bf16x8 cfrag = __builtin_bit_cast(bf16x8, up[0]);
    if constexpr (U == 0) {  // the buffer is only known to have landed after the barrier: cold start
      c1k[0] = ldsf(up + 1 * 64);
      c2r[0] = ldsf(up + 2 * 64);
      if constexpr (KS > 1) {
        c1k[1] = ldsf(up + 3 * 64);
        c2r[1] = ldsf(up + 4 * 64);
      }
    }
    auto fill = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      if constexpr (mi >= 2) {  // 8 register pairs of the previous unit's values over gaps 2 .. NM - 1
        constexpr int lo = ((mi - 2) * 16) / (NM - 2), hi = ((mi - 1) * 16) / (NM - 2);
        static_for<0, 8>([&](auto q_c) {
          constexpr int q = decltype(q_c)::value;
          if constexpr (2 * q + 1 >= lo && 2 * q + 1 < hi)
            take_keys_pair<2 * q>(b1[q & 1], b2[q & 1], fin[2 * q], fin[2 * q + 1]);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // small products first: corrections (C2 a1, C1 a2), main (C1 a1), then -N
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      if constexpr (st + 2 < KS) {
        c1k[st + 2] = ldsf(up + (1 + (st + 2) * 2) * 64);
        c2r[(st + 2) % 3] = ldsf(up + (2 + (st + 2) * 2) * 64);
      }
      if constexpr (st == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c2r[0], xs[0][0], zero, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c2r[st % 3], xs[st][0], acc, 0, 0, 0);
      }
      fill(std::integral_constant<int, 2 * st>{});
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1k[st], xs[st][1], acc, 0, 0, 0);
      fill(std::integral_constant<int, 2 * st + 1>{});
    });
    if constexpr (U < 3) {  // the next unit of the same buffer
      const u32x4* upn = up + FPU * 64;
      c2r[0] = ldsf(upn + 2 * 64);
      if constexpr (KS > 1) c2r[1] = ldsf(upn + (2 + 2) * 64);
    }
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1k[st], xs[st][0], acc, 0, 0, 0);
      if constexpr (st < 2 && U < 3) c1k[st] = ldsf(up + FPU * 64 + (1 + st * 2) * 64);
      fill(std::integral_constant<int, 2 * KS + st>{});
    });
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, acc, 0, 0, 0);
    fill(std::integral_constant<int, 3 * KS>{});
    bu[0] = b1[0] > before0 ? gprev : bu[0];
    bu[1] = b1[1] > before1 ? gprev : bu[1];
  };
  using std::integral_constant;
#pragma unroll 1
  for (int h = 0; h < n_half; ++h) {
    __syncthreads();  // half chunk h has landed (vmcnt(0) + barrier); everyone is done with the other buffer
    if (h + 1 < n_half) stage(h + 1);
    const u32x4* base = reinterpret_cast<const u32x4*>(smem + (h & 1) * HB) + lane;
    const int g = 4 * h;
    unit(integral_constant<int, 0>{}, base, accA, accB, g - 1);
    if (h == 0) {  // (the values processed under the very first unit were the -3e38 fill)
      b1[0] = b1[1] = b2[0] = b2[1] = -INFINITY;
      bu[0] = bu[1] = 0;
    }
    unit(integral_constant<int, 1>{}, base, accB, accA, g);
    unit(integral_constant<int, 2>{}, base, accA, accB, g + 1);
    unit(integral_constant<int, 3>{}, base, accB, accA, g + 2);
  }
  {  // the last unit's values
    const int glast = 4 * n_half - 1;
    const float before0 = b1[0], before1 = b1[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r >> 1) & 1;
      const float v = __int_as_float((__float_as_int(accB[r]) & ~15) | r);
      const float t = fminf(v, b1[c]);
      b1[c] = fmaxf(v, b1[c]);
      b2[c] = fmaxf(b2[c], t);
    }
    bu[0] = b1[0] > before0 ? glast : bu[0];
    bu[1] = b1[1] > before1 ? glast : bu[1];
  }
  const float s = a.scale[0];
  const float cn = sqrtf(__uint_as_float(a.cmax2_bits[0])), cnr = sqrtf(__uint_as_float(a.cmax2_bits[1]));
  const float c2n = -1.f;  // (level 2)
  const bool exact_all = (a.flag[0] | a.cflag[0]) != 0;
  const float inv_s2 = (1.f / s) * (1.f / s);
  {
    const int r0 = __float_as_int(b1[0]) & 15, r1 = __float_as_int(b1[1]) & 15;
    const int ia = bu[0] * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * half;
    const int ib = bu[1] * 32 + (r1 & 3) + 8 * (r1 >> 2) + 4 * half;
    const bool tb = b1[1] > b1[0] || (b1[1] == b1[0] && ib < ia);
    int idx = tb ? ib : ia;
    const float m1 = fmaxf(b1[0], b1[1]);
    const float m2 = fmaxf(fminf(b1[0], b1[1]), fmaxf(b2[0], b2[1]));
    const float o1 = __shfl_xor(m1, 32, 64), o2 = __shfl_xor(m2, 32, 64);
    const int oi = __shfl_xor(idx, 32, 64);
    const float B1 = fmaxf(m1, o1);
    const float B2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2));
    if (o1 > m1 || (o1 == m1 && oi < idx)) idx = oi;
    emit(a, bl, 0, lane, half == 0 && p >= 0, p, idx, B1, B2, n2, s, cn, cnr, inv_s2, exact_all, c2n);
  }
  flush_list(a, bl, 0);
}

// ---- chunked runs: fold the chunks and decide -----------------------------------------------------------
// One thread per point (level 1) or per list position (level 2): the chunks' (best, second, in-chunk
// index) in chunk order -- on a tie the earlier chunk, the smaller index, stays, and the tie itself makes
// second == best: the point is listed --, then the decision of emit().  grid (ceil(m / 256))
template <int LEVEL>
__global__ __launch_bounds__(256) void decide_kernel(StepArgs a, int n_chunks) {
  const int m = a.m;
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int cnt = m;
  if (LEVEL == 2) {
    cnt = a.count_in[0];
    cnt = cnt < m ? cnt : m;
    if ((int64_t)blockIdx.x * 256 >= cnt) return;  // block-uniform
  }
  const bool valid = pos < cnt;
  const int p = valid ? (LEVEL == 1 ? (int)pos : a.list_in[pos]) : 0;
  float B1 = -INFINITY, B2 = -INFINITY;
  int idx = 0;
  if (valid)
    for (int c = 0; c < n_chunks; ++c) {
      const float2 v = a.part_b[(int64_t)c * m + pos];
      const int i = a.part_i[(int64_t)c * m + pos];
      const float n2 = fmaxf(fminf(B1, v.x), fmaxf(B2, v.y));
      idx = v.x > B1 ? c * 256 + i : idx;
      B1 = fmaxf(B1, v.x);
      B2 = n2;
    }
  const float s = a.scale[0];
  const float cn = sqrtf(__uint_as_float(a.cmax2_bits[0])), cnr = sqrtf(__uint_as_float(a.cmax2_bits[1]));
  const float2 n2 = a.norms[p];
  float n2r, n2m;
  unpack_bound_norms(n2.y, n2r, n2m);
  const float an = sqrtf(n2.x), anr = sqrtf(n2r) * s;
  const float t1 = an + cn, t2 = anr + cnr * s;
  float dropped = 0.f;
  if (LEVEL == 1) {  // (emit())
    const float a2 = sqrtf(n2m), c2 = sqrtf(__uint_as_float(a.cmax2_bits[2]));
    dropped = a2 * (2.002f * cn + c2) + 1.001f * an * c2;
  }
  float delta = 1.25f * (dropped + a.eps * t1 * t1 + a.eta * (2.f * cn + an) + a.eps_exact * t2 * t2);
  if ((a.flag[0] | a.cflag[0]) != 0) delta = INFINITY;
  if (valid) {
    a.inds[p] = idx;
    if (a.vals) a.vals[p] = (B1 - n2.x) * ((1.f / s) * (1.f / s));
  }
  const bool listed = valid && !(B1 - B2 > 2.f * delta);
  const unsigned long long mk = __ballot(listed);
  if (mk) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mk) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(a.count, __popcll(mk));
    base = __shfl(base, leader, 64);
    if (listed) {
      const int slot = base + __popcll(mk & ((1ull << lane) - 1ull));
      a.list[slot] = p;
      // candidate route (cand_stream_kernel): every centroid at or above this may be the exact winner
      // (a flagged problem -- non-finite data, centroids beyond fp16's range: delta = inf, the keys may be inf or
      // NaN -- emits no candidates at all: gdecode_kernel sends its whole list to the exact kernel)
      if (LEVEL == 1 && a.thr && slot < a.thr_cap)
        a.thr[slot] = (a.flag[0] | a.cflag[0]) != 0
                          ? __builtin_nanf("")  // (no value compares >= NaN, not even inf)
                          : B1 - 2.f * delta - (fabsf(B1) * (1.0f / 65536.0f) + 1.0e-30f);
    }
  }
}

// ---- update from the pieces -------------------------------------------------------------------------
// sums[cluster][dim] = sum_i onehot(label_i)[cluster] * (h_i + m_i)[dim] on v_mfma_f32_32x32x16_f16 (exact
// products 1.0 * piece, fp32 sums: x is represented to 2^-22 relative, two ulps of fp32; the sums are
// of the centred, scaled values s (x - mu), undone by finalize).  Against centroid_accum_mfma_kernel
// (kmeans.hip) on the fp32 data: no splitting (it spends ~6.5 VALU instructions per element on three
// bf16 pieces), no LDS staging of the data, 64 + 8 MFMAs per 32 points x 64 dimensions instead of 96.
// The pieces are stored point-major (a lane of the assign kernels = one point, 8 dimensions); the
// update contracts over POINTS, so its B operand wants lane = dimension, 8 points.  The transposition
// runs on the matrix pipe: D = X I with the piece fragment as A (row = point, k = 16 dimensions)
// and an identity slice as B puts X[point][dim] into the accumulator layout -- lane = column = dimension,
// registers = rows = points -- exactly (one non-zero product per sum); eight v_cvt_pkrtz pack each
// 16 registers into the two k-steps' B fragments.  Point of k-slot i of k-group g in k-step s:
// 16 s + (i & 3) + 8 (i >> 2) + 4 g (the accumulator's row order) -- the one-hot operand uses the same.
// One wave owns 256 clusters x 32 dimensions (128 accumulator registers); the block's waves are the
// dimension tiles of the SAME points (their loads hit the same cache lines together).
struct UpdArgs {
  const u32x4* hi;
  const u32x4* mid;
  const int64_t* labels;  // [l][m]
  const int* flag;        // [l]: flagged sub-problems are skipped here (flagged_accum_kernel)
  float* sums;            // [l][d][k]
  float* counts;          // [l][k]
  int d, k, m;
  int64_t T;
};

// Wave (dt, ch) of a block owns dimensions [32 dt, 32 dt + 32) x clusters [128 ch, 128 ch + 128): 64
// accumulator registers, ~150 VGPRs in all -> three waves per SIMD.  (256 clusters per wave -- 128
// accumulator registers, two waves per SIMD -- ran 4.75 ms at C5: with one staging set a wave has one
// tile's loads in flight half of the time, 2048 waves x ~2 KiB, a third of what HBM latency x bandwidth
// asks for; a second or third staging set does not fit beside 128 accumulator registers -- 116 / 465
// spilled.  The four waves of a block read the same rows: one trip to HBM, the rest from L2 / L1.)
// NH = cluster halves per dimension tile (1: a wave owns all 256 clusters, 128 accumulator registers, two
// waves per SIMD; 2: 128 clusters, three waves per SIMD)
// (Round 5, same box, C5: one extra dword load per wave and tile touching the 32 cache lines of the tile two / four steps
// on -- to start their trip from HBM early, since a wave has one tile's fragments in flight -- made the update SLOWER:
// 3.85 -> 4.0 / 5.0 ms.  The kernel is not waiting on HBM latency: matrix pipe 41 % busy, 46 % of the wave cycles issuing,
// profiles/r04_c5.json; its time is the per-tile chain load -> transposing MFMAs -> pack -> one-hot rows through LDS -> MFMAs
// of two waves per SIMD, and more requests on the memory path lengthen it.)
template <int KS, int NH>
__global__ __launch_bounds__(64 * NH * ((KS + 1) / 2), (NH == 1 ? 2 : 3)) void update_kernel(UpdArgs a) {
  constexpr int DT = (KS + 1) / 2;
  constexpr int NW = NH * DT;
  constexpr int CW = 256 / NH, RT = 8 / NH;  // clusters / row tiles per wave
  __shared__ __attribute__((aligned(16))) uint16_t otab_s[NW][2 * CW * 8];  // [k-group][cluster][8 slots] fp16
  __shared__ int cnt_s[NW][CW];
  const int b = blockIdx.y;
  if (a.flag[b]) return;
  const int wave = threadIdx.x >> 6, dt = wave / NH, ch = wave % NH;
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  uint16_t* otab = otab_s[wave];
  int* cnt = cnt_s[wave];
  const int64_t slice = a.T * DT * 2048;
  const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.hi) + (size_t)b * slice), 0, (int)slice, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mid = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.mid) + (size_t)b * slice), 0, (int)slice, 0x00020000);
  const int64_t* __restrict__ lrow = a.labels + (int64_t)b * a.m;
  f32x16 acc[RT];  // [cluster row tile] x this wave's 32 dimensions
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
#pragma unroll
  for (int u = 0; u < CW / 64; ++u) cnt[lane + 64 * u] = 0;
#pragma unroll
  for (int u = 0; u < CW / 32; ++u) reinterpret_cast<u32x4*>(otab)[lane + 64 * u] = u32x4{0u, 0u, 0u, 0u};
  // identity slices: B[k][j] = (j == 16 ks2 + k), lane (j = l31, k-group half) holds k = 8 half .. + 7
  f16x8 ident[2];
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
    for (int i = 0; i < 8; ++i) ident[ks2][i] = (l31 == 16 * ks2 + 8 * half + i) ? (_Float16)1.0f : (_Float16)0.0f;
  // **The points of a tile are taken in the order of their clusters.**  A k-step of 16 random labels
  // touches 7 of the 8 cluster row tiles, so the one-hot multiply runs 8 x 2 MFMAs per k-step for 16
  // non-zero rows; with the tile's 32 points SORTED by label the first k-step holds the lower half of
  // the labels and the second the upper half -- 4 to 5 row tiles each -- and the row tiles outside
  // [first label, last label] of the k-step are skipped (wave-uniform branches): 4 + 2 x ~4.5 x 2 = 22
  // MFMAs per tile instead of 36.  The permutation costs nothing to apply: a lane loads the fragment of the
  // point of its SORTED position (the loads are per-lane addressed anyway), so the transposed pieces come
  // out in sorted order.  Cost: the labels of a tile are needed before its pieces (they are fetched three
  // steps ahead) and the grouping below, one step ahead, behind the MFMAs.
  constexpr unsigned kBad = 256u << 5;  // key of a point that is not this wave's (sorts last, row tile 8)
  // (grouping by cluster ROW TILE is all the skipping needs: a counting sort over the 9 row-tile values --
  // one ballot + two popcounts each -- and ONE ds_permute, instead of a 15-stage bitonic network whose
  // dependent cross-lane round trips cost more than the MFMAs they saved: 3.8 vs 3.5 ms at C5)
  auto sort32 = [&](unsigned key) -> unsigned {
    const unsigned rt = key >> 10;  // 0..7, 8 = not this wave's point
    const unsigned below_me = (1u << l31) - 1u;
    unsigned pos = 0, base = 0;
#pragma unroll
    for (unsigned r = 0; r <= 8; ++r) {
      const unsigned m32 = (unsigned)__ballot(rt == r);  // lanes 0..31 (32..63 hold the same points)
      pos = rt == r ? base + __popc(m32 & below_me) : pos;
      base += __popc(m32);
    }
    // lane `pos` (of the same half-wave) receives this lane's key
    return (unsigned)__builtin_amdgcn_ds_permute((int)((pos + 32 * half) * 4), (int)key);
  };
  // raw label of point l31 of `tile` (int64, clamped address: validity is applied when it is used)
  auto load_label = [&](int64_t tile) -> int64_t {
    const int64_t p = tile * 32 + l31;
    return lrow[(tile < a.T && p < a.m) ? p : 0];
  };
  auto key_of = [&](int64_t tile, int64_t label) -> unsigned {
    const int64_t p = tile * 32 + l31;
    const int64_t rel = label - CW * ch;  // this wave's clusters only: label - CW ch in [0, CW)
    const bool ok = tile < a.T && p < a.m && label < a.k && rel >= 0 && rel < CW;
    return ok ? (((unsigned)rel << 5) | (unsigned)l31) : (kBad | (unsigned)l31);
  };
  f16x8 x[2][2];  // [piece][k-step of the pair] of the tile in flight, rows in sorted order
  auto load_pieces = [&](int64_t tile, unsigned skey) {
    // lane (sorted position l31, half) takes the fragment of point (skey & 31) of the tile
    const int voff = tile < a.T ? (int)((tile * DT + dt) * 2048) + (int)(skey & 31u) * 64 + half * 16 : 0x7ffffff0;
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      x[0][ks2] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_hi, voff, ks2 * 32, 0));
      x[1][ks2] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_mid, voff, ks2 * 32, 0));
    }
  };
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // tiles dealt round-robin to the blocks of a sub-problem (adjacent rows are read side by side).  A
  // tile's fragments are consumed by its four transposing MFMAs right at the start; the loads of the
  // next tile go into the same registers immediately afterwards.
  const int64_t step = gridDim.x;
  unsigned skey = sort32(key_of(blockIdx.x, load_label(blockIdx.x)));  // sorted keys of the tile in flight
  load_pieces(blockIdx.x, skey);
  unsigned skey_n1 = sort32(key_of(blockIdx.x + step, load_label(blockIdx.x + step)));  // ... of the next tile
  int64_t lab_n2 = load_label(blockIdx.x + 2 * step);                  // raw labels of the tile after that
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < a.T; tile += step) {
    const unsigned rt_mine = skey >> 10;                       // cluster row tile of this lane's point (8: none)
    const int lab = skey < kBad ? (int)(skey >> 5) : -1;
    // transposition: [32 points][32 dims] of each piece into accumulator layout, packed at once into the
    // two k-steps' B fragments (one piece at a time: 16 transient registers)
    u32x4 bop[2][2];  // [piece][k-step of 16 points], fp16 pairs
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      f32x16 tr = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[pc][0], ident[0], zero, 0, 0, 0);
      tr = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[pc][1], ident[1], tr, 0, 0, 0);
#pragma unroll
      for (int sk = 0; sk < 2; ++sk)
#pragma unroll
        for (int i = 0; i < 4; ++i)  // (exact: every value IS an fp16 number)
          bop[pc][sk][i] = __builtin_bit_cast(
              uint32_t, __builtin_amdgcn_cvt_pkrtz(tr[8 * sk + 2 * i], tr[8 * sk + 2 * i + 1]));
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's pieces, in the order sorted during the previous step (the fragments are free now)
    load_pieces(tile + step, skey_n1);
    __builtin_amdgcn_sched_barrier(0);
    if (dt == 0 && half == 0 && lab >= 0) atomicAdd(&cnt[lab], 1);  // integer LDS atomic: fast
    // this lane's point (sorted position l31; half 0 lanes write): k-step l31 >> 4, k-group (l31 >> 2) & 1,
    // slot (l31 & 3) + 4 ((l31 >> 3) & 1)
    uint16_t* oslot = &otab[((l31 >> 2) & 1) * (CW * 8) + (lab >= 0 ? lab : 0) * 8 + (l31 & 3) + 4 * ((l31 >> 3) & 1)];
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
      // row tiles of this k-step's 16 (sorted) points: [lo, hi]; invalid points sort last (row tile 8)
      const int lo = (int)__builtin_amdgcn_readlane((int)rt_mine, 16 * sk);
      int hi = (int)__builtin_amdgcn_readlane((int)rt_mine, 16 * sk + 15);
      hi = hi < RT ? hi : RT - 1;
      const bool writer = half == 0 && (l31 >> 4) == sk && lab >= 0;
      if (writer) *oslot = (uint16_t)0x3C00;  // fp16 1.0
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const f16x8* orow = reinterpret_cast<const f16x8*>(&otab[half * (CW * 8) + l31 * 8]);
      f16x8 aop[RT];  // all A operands of the k-step are requested before its first MFMA
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (rt >= lo && rt <= hi) aop[rt] = orow[32 * rt];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (rt >= lo && rt <= hi) {
          acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop[rt], __builtin_bit_cast(f16x8, bop[0][sk]), acc[rt], 0, 0, 0);
          acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop[rt], __builtin_bit_cast(f16x8, bop[1][sk]), acc[rt], 0, 0, 0);
        }
      __builtin_amdgcn_wave_barrier();
      if (writer) *oslot = (uint16_t)0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // the tile two steps on: sort its labels (fetched a step ago) -- 15 dependent cross-lane stages, behind
    // this tile's MFMAs and in front of nothing: the pieces just requested have that long to land -- and
    // fetch the labels of the tile three steps on
    skey = skey_n1;
    skey_n1 = sort32(key_of(tile + 2 * step, lab_n2));
    lab_n2 = load_label(tile + 3 * step);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int dim = 32 * dt + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cluster = CW * ch + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float v = acc[rt][r];
      if (dim < a.d && cluster < a.k && v != 0.f) unsafeAtomicAdd(&a.sums[((int64_t)b * a.d + dim) * a.k + cluster], v);
    }
  }
  if (dt == 0) {
#pragma unroll
    for (int u = 0; u < CW / 64; ++u) {
      const int c = lane + 64 * u;
      if (CW * ch + c < a.k && cnt[c]) unsafeAtomicAdd(&a.counts[(int64_t)b * a.k + CW * ch + c], (float)cnt[c]);
    }
  }
}

// flagged sub-problems (data not finite / out of the fp16 range): raw fp32 sums straight to global atomics
// (each point confined to its own cluster, as compute_centroids.cu:10-86 has it).  grid (64, l): the
// blocks of an unflagged sub-problem leave at once (a grid over all points x dimensions was 16 M
// empty blocks: 3.7 ms)
__global__ __launch_bounds__(256) void flagged_accum_kernel(const float* __restrict__ data,
                                                           const int64_t* __restrict__ labels,
                                                           const int* __restrict__ flag, float* __restrict__ sums,
                                                           float* __restrict__ counts, int d, int m, int k) {
  const int b = blockIdx.y;
  if (!flag[b]) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const int64_t lab = labels[(int64_t)b * m + i];
    if (lab < 0 || lab >= k) continue;
    unsafeAtomicAdd(&counts[(int64_t)b * k + lab], 1.0f);
    for (int e = 0; e < d; ++e)
      unsafeAtomicAdd(&sums[((int64_t)b * d + e) * k + lab], data[((int64_t)b * d + e) * m + i]);
  }
}

// centroid = mu + (sums / count) / s (raw sums / count for flagged sub-problems); empty cluster -> 0
__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ sums, const float* __restrict__ counts,
                                                      const float* __restrict__ mu, const float* __restrict__ scale,
                                                      const int* __restrict__ flag, float* __restrict__ out, int d,
                                                      int k, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % k);
  const int dim = (int)((t / k) % d);
  const int64_t b = t / ((int64_t)d * k);
  const float cnt = counts[b * k + c];
  float v = 0.f;  // compute_centroids.cu:82
  if (cnt != 0.f) v = flag[b] ? sums[t] / cnt : mu[b * kMu + dim] + (sums[t] / cnt) * (1.f / scale[b]);
  out[t] = v;
}

template <int KS>
static int run_update(const UpdArgs& ua, const float* data, const float* mu, const float* scale, float* out, int l,
                      hipStream_t st) {
  constexpr int DT = (KS + 1) / 2;
  // one round of resident waves: every wave ends with its (256 / NH) x 32 global atomics
  const char* e = TPQ_AB_ENV("TPQ_LL_NH");  // (variant builds, A/B)
  const int nh = e ? atoi(e) : 1;
  int64_t chunks = (nh == 1 ? 2048 : 3072) / ((int64_t)l * nh * DT);
  if (chunks < 1) chunks = 1;
  if (chunks > ua.T) chunks = ua.T;
  if (nh == 1)
    hipLaunchKernelGGL((update_kernel<KS, 1>), dim3((unsigned)chunks, l), dim3(64 * DT), 0, st, ua);
  else
    hipLaunchKernelGGL((update_kernel<KS, 2>), dim3((unsigned)chunks, l), dim3(128 * DT), 0, st, ua);
  TPQ_LAUNCH_CHECK("lloyd update_kernel");
  hipLaunchKernelGGL(flagged_accum_kernel, dim3(64, l), dim3(256), 0, st, data,
                     ua.labels, ua.flag, ua.sums, ua.counts, ua.d, ua.m, ua.k);
  TPQ_LAUNCH_CHECK("lloyd flagged_accum_kernel");
  const int64_t total = (int64_t)l * ua.d * ua.k;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ua.sums, ua.counts, mu,
                     scale, ua.flag, out, ua.d, ua.k, total);
  TPQ_LAUNCH_CHECK("lloyd finalize_kernel");
  return TPQ_OK;
}

// ---- step workspace --------------------------------------------------------------------------------
struct StepLayout {
  size_t frags_off, cmax_off, count_off, cflag_off, count2_off, list_off, list2_off, upd_off, total;
};
static StepLayout step_layout(int l, int d, int64_t m, int n) {
  StepLayout L;
  const int KS = ks_of(d);
  L.frags_off = 0;
  L.cmax_off = (size_t)l * 8 * (2 * KS + 1) * 1024;   // [l][kCm] u32
  L.count_off = L.cmax_off + (size_t)l * 4 * kCm;      // [l] i32: left undecided by level 1
  L.cflag_off = L.count_off + (size_t)l * 4;           // [l] i32
  L.count2_off = L.cflag_off + (size_t)l * 4;          // [l] i32: left undecided by level 2
  L.list_off = (L.count2_off + (size_t)l * 4 + 255) / 256 * 256;   // [l][m] i32
  L.list2_off = (L.list_off + (size_t)l * m * 4 + 255) / 256 * 256;  // [l][m] i32
  L.upd_off = (L.list2_off + (size_t)l * m * 4 + 255) / 256 * 256;
  L.total = L.upd_off + tpq_compute_centroids_workspace_bytes(l, d, n);
  return L;
}

// the fast-path bounds of level 1 / level 2, relative to (|a'| + |c'|max)^2
static float level_eps(int KS, int d, int level) {
  const int terms = KS * 16 + 2 + 3;
  const float common = (float)(terms + 8) / 8388608.0f + (float)(d + 1) / 16777216.0f + 1.0f / 4194304.0f +
                       1.0f / 524288.0f;  // accumulation, norm chain, shift rounding, key bits
  return level == 1 ? common + 1.0f / 131072.0f  // (the dropped pieces: per point, emit()); 6-bit keys: 2^-17
                    : 3.03f / 4194304.0f + common;
}

template <int KS>
static int run_levels(StepArgs sa, int l, int d, int* list2, int* count2, hipStream_t st) {
  {  // level 1
    const size_t lds = (size_t)8 * (KS + 1) * 1024 + sizeof(BlockListT<kCoarseList>);
    auto kernel = coarse_kernel<KS>;
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "lloyd coarse_kernel attr");
    if (rc) return rc;
    sa.eps = level_eps(KS, d, 1);
    sa.level = 1;
    const int64_t wide = (sa.T + 1) / 2, per_block = (int64_t)kWaves * kWide;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((wide + per_block - 1) / per_block), l), dim3(kWaves * 64), lds, st, sa);
    TPQ_LAUNCH_CHECK("lloyd coarse_kernel");
  }
  {  // level 2
    const size_t lds = (size_t)8 * (2 * KS + 1) * 1024 + sizeof(BlockListT<kRefineList>);
    auto kernel = refine_kernel<KS>;
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "lloyd refine_kernel attr");
    if (rc) return rc;
    sa.eps = level_eps(KS, d, 2);
    sa.level = 2;
    sa.list_in = sa.list;
    sa.count_in = sa.count;
    sa.list = list2;
    sa.count = count2;
    const int64_t per_block = (int64_t)kWaves * kTilesR;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((sa.T + per_block - 1) / per_block), l), dim3(kWaves * 64), lds, st, sa);
    TPQ_LAUNCH_CHECK("lloyd refine_kernel");
  }
  return TPQ_OK;
}


// ---- many centroids: tpq_coarse_assign through the cascade (one problem, d <= 128) --------------------
// The centroids in chunks of 256 = blockIdx.y of ONE launch per level (a chunk's blocks alone fill half
// the chip); a chunk's (best, second, in-chunk index) of every point goes to part_*, decide_kernel folds
// the chunks and decides.  Level 1 = coarse_kernel over the points' hi pieces, level 2 = refine_kernel
// over the undecided points, level 3 = the exact kernel over what is left (compact copy + centroid
// splits, as in assign_fast.hip).  16 GB of piece streaming at 1 M x 16 384 x 128 instead of the
// three-product sweep of assign_fast_kernel.
struct AssignLayout {
  PrepLayout P;
  int KS, chunks, cap;
  size_t prep_off, frags_off, cmax_off, cflag_off, count1_off, count2_off, partb_off, parti_off, list1_off, list2_off,
      keys_off, ac_off, total;
  // candidate route (chunked problems): thresholds, pairs, row copies
  int cap2, pair_cap, dp;
  size_t npairs_off, oflag_off, countfb_off, thr_off, pairs_off, bt_off, xt_off;
};
static AssignLayout assign_layout(int d, int64_t m, int n) {
  AssignLayout L;
  L.P = prep_layout(1, d, m);
  L.KS = L.P.KS;
  L.chunks = (n + 255) / 256;
  L.cap = (int)((m / 4 + 127) / 128 * 128);
  if (L.cap < 128) L.cap = 128;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  L.prep_off = 0;
  L.frags_off = up(L.P.total);
  L.cmax_off = up(L.frags_off + (size_t)L.chunks * 8 * (2 * L.KS + 1) * 1024);
  L.cflag_off = L.cmax_off + 4 * kCm;
  L.count1_off = L.cflag_off + 4;
  L.count2_off = L.count1_off + 4;
  L.npairs_off = L.count2_off + 4;
  L.oflag_off = L.npairs_off + 4;
  L.countfb_off = L.oflag_off + 4;
  L.partb_off = up(L.countfb_off + 4);                              // [chunks][m] float2
  L.parti_off = up(L.partb_off + (size_t)L.chunks * m * 8);         // [chunks][m] u8
  L.list1_off = up(L.parti_off + (size_t)L.chunks * m);
  L.list2_off = up(L.list1_off + (size_t)m * 4);
  L.keys_off = up(L.list2_off + (size_t)m * 4);        // [m] u64 (level 3: zeroed)
  L.ac_off = up(L.keys_off + (size_t)m * 8);           // [d][cap] f32
  L.total = up(L.ac_off + (size_t)d * L.cap * 4);
  // candidate route: the listed points' thresholds and row copies (a quarter of the points), the pairs, the
  // centroids as rows
  L.cap2 = (int)(((m / 4 > 8192 ? m / 4 : 8192) + 511) / 512 * 512);
  if ((int64_t)L.cap2 > (m + 511) / 512 * 512) L.cap2 = (int)((m + 511) / 512 * 512);
  L.pair_cap = 4 * L.cap2 > 65536 ? 4 * L.cap2 : 65536;
  L.dp = (d + 15) / 16 * 16;
  L.thr_off = L.total;
  L.pairs_off = up(L.thr_off + (size_t)L.cap2 * 4);
  L.bt_off = up(L.pairs_off + (size_t)L.pair_cap * 8);
  L.xt_off = up(L.bt_off + (size_t)n * L.dp * 4);
  L.total = up(L.xt_off + (size_t)L.cap2 * L.dp * 4);
  return L;
}

// the candidate route of a chunked problem (defined behind the wide path, whose pair machinery it shares)
template <int KS>
static int run_cand_tail(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n, char* ws,
                         const AssignLayout& L, const u32x4* hi, const u32x4* frags, hipStream_t st);

template <int KS>
static int run_assign(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n, char* ws,
                      const AssignLayout& L, hipStream_t st) {
  const PrepLayout& P = L.P;
  char* p = ws + L.prep_off;
  float* mu = reinterpret_cast<float*>(p + P.mu_off);
  float* scale = reinterpret_cast<float*>(p + P.scale_off);
  int* flag = reinterpret_cast<int*>(p + P.flag_off);
  unsigned* maxbits = reinterpret_cast<unsigned*>(p + P.maxbits_off);
  u32x4* frags = reinterpret_cast<u32x4*>(ws + L.frags_off);
  unsigned* cmax = reinterpret_cast<unsigned*>(ws + L.cmax_off);
  int* cflag = reinterpret_cast<int*>(ws + L.cflag_off);
  int* count1 = reinterpret_cast<int*>(ws + L.count1_off);
  int* count2 = reinterpret_cast<int*>(ws + L.count2_off);
  float2* part_b = reinterpret_cast<float2*>(ws + L.partb_off);
  uint8_t* part_i = reinterpret_cast<uint8_t*>(ws + L.parti_off);
  int* list1 = reinterpret_cast<int*>(ws + L.list1_off);
  int* list2 = reinterpret_cast<int*>(ws + L.list2_off);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + L.keys_off);
  float* Ac = reinterpret_cast<float*>(ws + L.ac_off);
  int rc = check_hip(hipMemsetAsync(p + P.mu_off, 0, P.total - P.mu_off, st), "coarse_assign memset");
  if (rc) return rc;
  rc = check_hip(hipMemsetAsync(ws + L.cmax_off, 0, L.partb_off - L.cmax_off, st), "coarse_assign memset");
  if (rc) return rc;
  rc = check_hip(hipMemsetAsync(keys, 0, (size_t)m * 8, st), "coarse_assign keys memset");
  if (rc) return rc;
  // prepare the points (per call: the points change from call to call, the centroids are the codebook)
  hipLaunchKernelGGL(mu_kernel, dim3(d, 1), dim3(256), 0, st, B, mu, d, n);
  TPQ_LAUNCH_CHECK("lloyd mu_kernel");
  int chunks = (int)(4096 / (int64_t)d);
  if (chunks < 1) chunks = 1;
  if ((int64_t)chunks * 4096 > m) chunks = (int)((m + 4095) / 4096);
  hipLaunchKernelGGL(maxabs_kernel, dim3(chunks, d, 1), dim3(256), 0, st, A, mu, maxbits, flag, d, m);
  TPQ_LAUNCH_CHECK("lloyd maxabs_kernel");
  hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(64), 0, st, maxbits, flag, scale, 1);
  TPQ_LAUNCH_CHECK("lloyd scale_kernel");
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((P.T + 7) / 8), 1), dim3(256), 0, st, A, mu, scale,
                     reinterpret_cast<u32x4*>(p + P.hi_off), reinterpret_cast<u32x4*>(p + P.mid_off),
                     reinterpret_cast<float2*>(p + P.norms_off), flag, d, m, P.T, KS);
  TPQ_LAUNCH_CHECK("lloyd split_kernel");
  hipLaunchKernelGGL(cprep_kernel, dim3(8 * L.chunks, 1), dim3(64), 0, st, B, mu, scale, frags, cmax, cflag, d, n, KS);
  TPQ_LAUNCH_CHECK("lloyd cprep_kernel");
  const bool chunked = L.chunks > 1;
  const int chunk_stride = 8 * (2 * KS + 1) * 64;  // 16-byte units
  StepArgs sa{reinterpret_cast<const u32x4*>(p + P.hi_off), reinterpret_cast<const u32x4*>(p + P.mid_off),
              reinterpret_cast<const float2*>(p + P.norms_off), frags, cmax, scale, flag, cflag, inds, vals,
              nullptr, nullptr, list1, count1, (int)m, P.T,
              level_eps(KS, d, 1), (float)(d + 4) / 16777216.0f, sqrtf((float)d) / 8192.0f, 1, nullptr, 0,
              chunked ? part_b : nullptr, chunked ? part_i : nullptr, chunk_stride};
  {
    const size_t lds = (size_t)8 * (KS + 1) * 1024 + sizeof(BlockListT<kCoarseList>);
    auto kernel = coarse_kernel<KS>;
    rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "lloyd coarse_kernel attr");
    if (rc) return rc;
    const int64_t wide = (P.T + 1) / 2, per_block = (int64_t)kWaves * kWide;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((wide + per_block - 1) / per_block), L.chunks), dim3(kWaves * 64), lds,
                       st, sa);
    TPQ_LAUNCH_CHECK("lloyd coarse_kernel");
    // chunked problems: the undecided points do not go through levels 2 and 3 -- their CANDIDATES (the centroids
    // within twice level 1's bound of the best: ~2 per point) get the exact kernel's value (the wide path's
    // machinery).  TPQ_COARSE_ASSIGN_CAND=0: levels 2 and 3 (A/B)
    static const bool cand_route = !(TPQ_AB_ENV("TPQ_COARSE_ASSIGN_CAND") && atoi(TPQ_AB_ENV("TPQ_COARSE_ASSIGN_CAND")) == 0);
    if (chunked) {
      const bool cand = cand_route && n <= (1 << 22);  // (a pair entry carries the centroid in 22 bits)
      if (cand) {
        sa.thr = reinterpret_cast<float*>(ws + L.thr_off);
        sa.thr_cap = L.cap2;
      }
      hipLaunchKernelGGL(decide_kernel<1>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, sa, L.chunks);
      TPQ_LAUNCH_CHECK("lloyd decide_kernel");
      if (cand)
        return run_cand_tail<KS>(A, B, vals, inds, d, m, n, ws, L, reinterpret_cast<const u32x4*>(p + P.hi_off), frags, st);
    }
  }
  {
    const size_t lds = (size_t)8 * (2 * KS + 1) * 1024 + sizeof(BlockListT<kRefineList>);
    auto kernel = refine_kernel<KS>;
    rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "lloyd refine_kernel attr");
    if (rc) return rc;
    sa.eps = level_eps(KS, d, 2);
    sa.level = 2;
    sa.list_in = list1;
    sa.count_in = count1;
    sa.list = list2;
    sa.count = count2;
    if (chunked) {  // the points stay in registers, the chunks stream
      sa.part_b = nullptr;
      sa.part_i = nullptr;
      const size_t lds2 = (size_t)2 * 4 * (2 * KS + 1) * 1024 + sizeof(BlockListT<kStreamList>);
      auto k2 = refine_stream_kernel<KS>;
      rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds2), "lloyd refine_stream_kernel attr");
      if (rc) return rc;
      hipLaunchKernelGGL(k2, dim3((unsigned)((m + kWaves * 32 - 1) / (kWaves * 32))), dim3(kWaves * 64), lds2, st, sa,
                         2 * L.chunks);
      TPQ_LAUNCH_CHECK("lloyd refine_stream_kernel");
    } else {
      const int64_t per_block = (int64_t)kWaves * kTilesR;
      hipLaunchKernelGGL(kernel, dim3((unsigned)((P.T + per_block - 1) / per_block), 1), dim3(kWaves * 64), lds, st, sa);
      TPQ_LAUNCH_CHECK("lloyd refine_kernel");
    }
  }
  return launch_max_sim_list(A, B, vals, inds, 1, d, (int)m, n, 1, list2, count2, keys, Ac, L.cap, st);
}

// =========================================================================================================
// Wide vectors (128 < d <= 1024): GEMM-shaped, and the exact step works on CANDIDATES (the
// coarse assign of a GIST-dimension index -- 1 M x 16 384 x 960 -- is 270-320 ms on the fp32 MFMA)
// =========================================================================================================
// Beyond 128 dimensions a wave can no longer keep its points' pieces in registers, so both operands are
// streamed: a block owns 256 points x ALL centroids; per block of 256 centroids it runs the K loop over the
// k-steps with both operands' fragments coming through a double-buffered LDS ring by LDS-DMA (4 k-steps per
// stage: 32 KiB of centroid fragments + 32 KiB of point fragments), 8 waves as 4 point slabs x 2 centroid
// halves, each wave 64 points x 128 centroids = 8 accumulators; after the K loop the -N MFMAs and the
// epilogue over the 128 values per lane -- amortised over KS x 8 MFMAs.
//
// At d ~ 1000 every (d 2^-24)-sized term of the bound -- the fast path's accumulation AND the exact kernel's
// own rounding, which no fast path can remove -- is worth percents of the points: a third product pass
// (level 2 of the narrow path) would turn 18 % undecided into 11 %, and the exact kernel over 11 % of a
// million points is a third of the full problem.  So the wide path never runs the exact kernel over all
// centroids.  What the bound does give for an undecided point is a short list of CANDIDATES: the exact
// winner w satisfies f_w >= f_best - 2 delta.  Three passes:
//   1. gemm_kernel<false>: hi pieces, key top-2 of every point; gdecide_kernel decides 80 % and lists the rest
//      with its threshold f_best - 2 delta;
//   2. gemm_kernel<true> over the listed points (hi pieces gathered into a compact fragment array): every
//      (point, centroid) at or above the point's threshold goes to a pair list -- 2-3 per listed point;
//   3. pair_exact_kernel: the exact kernel's value of each pair -- the SAME instruction sequence as
//      max_sim_kernel (kmeans.hip): ascending-k fma chains for |x|^2 and |c|^2, v_mfma_f32_32x32x2f32 over
//      ascending k pairs, 2 acc - |x|^2 - |c|^2 -- 32 pairs per wave on the diagonal of a 32 x 32 tile,
//      folded per point with the 64-bit atomicMax key of the exact kernel's split mode (value, then the
//      smaller index).
// Pair lists that overflow (degenerate data: duplicates by the thousand, non-finite values) fall back to
// the exact kernel over the level-1 list: correctness never depends on the candidate counts.
// Operand layout: plain MFMA-fragment order [tile of 32 rows][k-step][lane] x 16 B for points and
// centroids alike (every access is a 1-KiB stream).
constexpr int kGK = 4;                       // k-steps per LDS stage
constexpr int kGStage = 2 * 8 * kGK * 1024;  // bytes of a stage: 8 centroid units + 8 point tiles, kGK k-steps each
constexpr int kPairList = 4096;              // LDS-staged (row, centroid) entries per block between flushes

struct PairList {
  int n;
  int base;
  unsigned item[kPairList];  // row in the block (8 bits) << 22 | centroid
};

struct GemmArgs {
  const u32x4* cfr;      // centroid operand [units][KA][64]
  const u32x4* pfr;      // point operand [tiles][KA][64]
  const u32x4* cnorm;    // [units][64]: -N as three bf16 pieces at k = 0, 1, 2 (-3e38 beyond n)
  float2* part_b;        // pass 1: [2 * ysplit][rows] (best, second) per centroid half
  int* part_i;           // pass 1: [2 * ysplit][rows] centroid index
  int KA;                // k-steps of the operands (multiple of kGK)
  int n_cblocks;         // blocks of 256 centroids
  int ysplit;            // 1, 2, 4 or 8 ranges of centroid blocks (see the block mapping in the kernel)
  int pblocks;           // blocks of 256 points the grid covers
  int64_t rows;          // points (pass 1) / capacity of the compact array (pass 2)
  const int* count_in;   // pass 2: number of listed points on the device (rows = min(this, capacity))
  const float* thr;      // pass 2: [capacity] threshold of the listed point
  uint2* pairs;          // pass 2: (list position, centroid)
  int* pair_count;
  int pair_cap;
  int* overflow;         // pass 2: set when a pair was dropped
  int n_centroids;       // pass 2: centroids beyond are padding (never a pair, whatever their value)
};

// the block's staged pairs -> the global list.  Called by all threads, at points where nobody appends.
__device__ __forceinline__ void flush_pairs(const GemmArgs& a, PairList* pl, unsigned row0) {
  if (threadIdx.x == 0) {
    const int n = pl->n < kPairList ? pl->n : kPairList;
    if (pl->n > kPairList) atomicOr(a.overflow, 1);
    pl->base = n ? atomicAdd(a.pair_count, n) : 0;
  }
  __syncthreads();
  const int n = pl->n < kPairList ? pl->n : kPairList, base = pl->base;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned e = pl->item[i];
    if (base + i < a.pair_cap)
      a.pairs[base + i] = make_uint2(row0 + (e >> 22), e & 0x3fffffu);
    else
      atomicOr(a.overflow, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) pl->n = 0;
}

// CT = point column tiles per wave: 2 -> 8 waves as 4 point slabs x 2 centroid halves (wave tile 128 x 64),
// 4 -> 4 waves as 2 x 2 (wave tile 128 x 128, one wave per SIMD, 16 accumulators)
template <bool CAND, int CT>
__global__ __launch_bounds__(CT == 2 ? 512 : 256) void gemm_kernel(GemmArgs a) {
  constexpr int NW = CT == 2 ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages, the block's -N fragments (8 KiB), PairList
  int64_t rows = a.rows;
  if (CAND) {
    const int c = *a.count_in;
    rows = c < a.rows ? c : a.rows;
  }
  // Block -> (point block pb, centroid range yi), XCD-aware.  A 256 x 256 tile streams 2 x 480 KiB at d = 960 and
  // every point block meets every centroid block: with one point block per CU walking all centroids, the 32
  // CUs of an XCD share the centroid stream but re-read 32 x 480 KiB = 15 MiB of point fragments per step --
  // four times their 4-MiB L2, i.e. from the Infinity Cache every time.  So the 32 blocks an XCD runs
  // side by side (the dispatcher deals block b to XCD b % 8, in order) are P = 32 / ysplit point blocks x
  // ysplit centroid ranges: per step they stream P + ysplit tiles instead of 33 (ysplit = 8: 12).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, ys = a.ysplit, P = 32 / ys;
  const int pb = ((slot >> 5) * 8 + xcd) * P + (slot & 31) / ys, yi = (slot & 31) % ys;
  if (pb >= a.pblocks || (int64_t)pb * 256 >= rows) return;  // block-uniform
  // (the wave number as a SCALAR: everything derived from it -- the addresses of the LDS-DMA fills above all --
  // is then SALU work; as a vector it was 6 VALU instructions per MFMA in the issue slots the MFMAs need)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;  // point slab (64 points), centroid half (128 centroids)
  const int cb_per = (a.n_cblocks + ys - 1) / ys;
  const int cb0 = yi * cb_per, cb1 = (cb0 + cb_per) < a.n_cblocks ? (cb0 + cb_per) : a.n_cblocks;
  if (cb0 >= cb1) return;
  const int n_kst = a.KA / kGK;  // stages per centroid block (>= 3: d > 128)
  const int n_stage = (cb1 - cb0) * n_kst;
  char* nbuf = smem + 2 * kGStage;
  PairList* pl = reinterpret_cast<PairList*>(nbuf + 8 * 1024);
  if (CAND && threadIdx.x == 0) pl->n = 0;
  // stage (centroid block cb, k-steps [kGK kst, + kGK)) -> buffer buf: units 0..7 then point tiles 0..7, each
  // kGK consecutive 1-KiB fragments
  const unsigned lane16 = lane * 16;
  auto stage = [&](int cb, int kst, int buf) {
    char* dst = smem + buf * kGStage;
    const char* csrc = reinterpret_cast<const char*>(a.cfr) + ((size_t)cb * 8 * a.KA + kst * kGK) * 1024;
    const char* psrc = reinterpret_cast<const char*>(a.pfr) + ((size_t)pb * 8 * a.KA + kst * kGK) * 1024;
#pragma unroll
    for (int i = 0; i < 8 * kGK / NW; ++i) {  // this wave's share of the 8 kGK centroid fragments, then of the points'
      const int f = wave + i * NW, who = f / kGK, kk = f % kGK;
      const size_t off = ((size_t)who * a.KA + kk) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(csrc + off + lane16),
                                       (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(psrc + off + lane16),
                                       (__attribute__((address_space(3))) void*)(dst + (8 * kGK + f) * 1024), 16, 0, 0);
    }
  };
  stage(cb0, 0, 0);
  f32x16 acc[4][CT];  // [centroid row tile of this wave's half][point column tile of its slab]
  float b1[CT], b2[CT], thr[CT];
  int bu[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    b1[ct] = -INFINITY;
    b2[ct] = -INFINITY;
    bu[ct] = 0;
    // (NaN: no value compares >= it.  With +inf a column of garbage fragments -- the rows of the last block beyond
    // the list are never gathered -- that happened to hold an inf passed the test: a pair of a list position
    // beyond the count, an address from an unwritten list entry, a memory fault: tools/selection_soak.py seed 11)
    thr[ct] = __builtin_nanf("");
    if (CAND) {
      const int64_t row = (int64_t)pb * 256 + wr * (CT * 32) + ct * 32 + l31;
      if (row < rows) thr[ct] = a.thr[row];
    }
  }
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int cb = cb0, kst = 0;
#pragma unroll 1
  for (int g = 0; g < n_stage; ++g, ++kst) {
    if (kst == n_kst) {
      kst = 0;
      ++cb;
    }
    __syncthreads();  // stage g has landed; everyone is done with the other buffer (and with nbuf)
#ifndef TPQ_EXP_NODMA  // (experiment: the K loop without its LDS-DMA fills, garbage results: 22.6 instead of 28.7 ms.
                       // Two restructurings that were built and measured, neither kept: the points' fragments global ->
                       // registers instead (half the fills, a third fewer LDS reads): 30.4 ms -- fragment-shaped
                       // loads cost more on the vector memory path than they save in LDS; a ring of four 2-k-step
                       // stages with inline-asm fills and a barrier that leaves the newest stage in flight
                       // (`s_waitcnt vmcnt(4)`), next stage's first fragments read before the barrier: 31.0 ms --
                       // twice the barriers cost more than the refill bubble they remove)
    if (g + 1 < n_stage) {
      const bool wrap = kst + 1 == n_kst;
      stage(wrap ? cb + 1 : cb, wrap ? 0 : kst + 1, (g + 1) & 1);
    }
#endif
    if (kst == 0) {  // a new centroid block: its -N fragments (8 KiB; read after the K loop) and fresh accumulators
      for (int f = wave; f < 8; f += NW)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(a.cnorm) +
                                                           ((size_t)cb * 8 + f) * 1024 + lane * 16),
            (__attribute__((address_space(3))) void*)(nbuf + f * 1024), 16, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = zero;
      // (pairs are appended in the epilogue of the previous centroid block, before the barrier above:
      // pl->n is stable here, and the condition uniform)
      if (CAND && pl->n >= kPairList / 2) flush_pairs(a, pl, pb * 256u);
    }
    const u32x4* sb = reinterpret_cast<const u32x4*>(smem + (g & 1) * kGStage) + lane;
    // software pipeline inside the stage: the 4 + CT fragments of k-step kk + 1 are read before the MFMAs of
    // k-step kk issue (left to itself the compiler reads ONE A fragment, waits, issues its CT MFMAs, reads the next)
    f16x8 af[2][4], bf[2][CT];
    auto read_frags = [&](int kk, int set) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) af[set][rt] = __builtin_bit_cast(f16x8, sb[((wc * 4 + rt) * kGK + kk) * 64]);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        bf[set][ct] = __builtin_bit_cast(f16x8, sb[((8 + wr * CT + ct) * kGK + kk) * 64]);
    };
    read_frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < kGK; ++kk) {
      if (kk + 1 < kGK) read_frags(kk + 1, (kk + 1) & 1);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk & 1][rt], bf[kk & 1][ct], acc[rt][ct], 0, 0, 0);
      if (kk + 1 < kGK) __builtin_amdgcn_sched_group_barrier(0x100, 4 + CT, 0);  // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * CT, 0);                      // MFMA
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kst == n_kst - 1) {  // the centroid block is complete: -N, then the epilogue over CT x 64 values per lane
      // (nbuf was requested n_kst stages ago and every barrier since waited for vmcnt(0))
      const u32x4* nb = reinterpret_cast<const u32x4*>(nbuf) + lane;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const bf16x8 cf = __builtin_bit_cast(bf16x8, nb[(wc * 4 + rt) * 64]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf, bones, acc[rt][ct], 0, 0, 0);
      }
      if (!CAND) {
        float before[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) before[ct] = b1[ct];
        static_for<0, 4>([&](auto rt_c) {
          constexpr int rt = decltype(rt_c)::value;
          static_for<0, 8 * CT>([&](auto q_c) {  // 8 register pairs x CT column tiles
            constexpr int q = decltype(q_c)::value, ct = q % CT, pq = q / CT;
            top2_keys_pair(b1[ct], b2[ct], key6<16 * rt + 2 * pq>(acc[rt][ct][2 * pq]),
                           key6<16 * rt + 2 * pq + 1>(acc[rt][ct][2 * pq + 1]));
          });
        });
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bu[ct] = b1[ct] > before[ct] ? cb : bu[ct];
      } else {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            float mx = acc[rt][ct][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[rt][ct][r]);
            if (__ballot(mx >= thr[ct]) == 0ull) continue;  // (nearly every tile)
            const unsigned rowbits = (unsigned)(wr * (CT * 32) + ct * 32 + l31) << 22;
            const int cbase = cb * 256 + wc * 128 + rt * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool hit = acc[rt][ct][r] >= thr[ct] && cbase + (r & 3) + 8 * (r >> 2) < a.n_centroids;
              const unsigned long long mk = __ballot(hit);
              if (mk) {
                const int leader = __ffsll((long long)mk) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&pl->n, __popcll(mk));  // LDS
                base = __shfl(base, leader, 64);
                const int slot = base + __popcll(mk & ((1ull << lane) - 1ull));
                if (hit && slot < kPairList) pl->item[slot] = rowbits | (unsigned)(cbase + (r & 3) + 8 * (r >> 2));
              }
            }
          }
      }
    }
  }
  if (CAND) {
    __syncthreads();
    flush_pairs(a, pl, pb * 256u);
    return;
  }
  // this wave's (best, second, index) of its 2 x 32 points over its centroid half of every block
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int tag = __float_as_int(b1[ct]) & 63, r0 = tag & 15;
    int idx = bu[ct] * 256 + wc * 128 + (tag >> 4) * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * half;
    const float m1 = b1[ct], m2 = b2[ct];
    const float o1 = __shfl_xor(m1, 32, 64), o2 = __shfl_xor(m2, 32, 64);
    const int oi = __shfl_xor(idx, 32, 64);
    const float B1 = fmaxf(m1, o1);
    const float B2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2));
    if (o1 > m1 || (o1 == m1 && oi < idx)) idx = oi;
    const int64_t row = (int64_t)pb * 256 + wr * (CT * 32) + ct * 32 + l31;
    if (half == 0 && row < rows) {
      const int64_t slot = (int64_t)(yi * 2 + wc) * a.rows + row;
      a.part_b[slot] = make_float2(B1, B2);
      a.part_i[slot] = idx;
    }
  }
}

// points -> fragment order: hi [T][KAp][64] x 16 B (k-steps beyond ceil(d / 16) zero), norms.  grid (ceil(T / 4))
__global__ __launch_bounds__(256) void gsplit_points_kernel(const float* __restrict__ A, const float* __restrict__ mu,
                                                           const float* __restrict__ scale, u32x4* __restrict__ hi,
                                                           float4* __restrict__ norms, int d, int64_t m, int64_t T,
                                                           int KAp) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= T) return;
  const int64_t i = tile * 32 + l31;
  const bool iv = i < m;
  const float* Ap = A + (iv ? i : 0);
  const float s = scale[0];
  float n2c = 0.f, n2r = 0.f, n2m = 0.f;
  for (int st = 0; st < KAp; ++st) {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * st + 8 * half + j;
      const float x = (iv && k < d) ? Ap[(int64_t)k * m] : 0.f;
      const float v = (iv && k < d) ? (x - mu[k]) * s : 0.f;
      const _Float16 hh = (_Float16)v;
      const float r = v - (float)hh;  // exact: what the fast path drops of this element
      h[j] = hh;
      n2c = fmaf(v, v, n2c);
      n2r = fmaf(x, x, n2r);
      n2m = fmaf(r, r, n2m);
    }
    hi[(tile * KAp + st) * 64 + lane] = __builtin_bit_cast(u32x4, h);
  }
  n2c += __shfl_xor(n2c, 32, 64);
  n2r += __shfl_xor(n2r, 32, 64);
  n2m += __shfl_xor(n2m, 32, 64);
  if (half == 0 && iv) norms[i] = make_float4(n2c, n2r, n2m, 0.f);
}

// centroids -> operand c1 [U][KAp][64] (hi piece of C = 2 c'; inner product: C = c'), -N fragments, max
// norms, range flag.  grid (U), U = 8 x blocks of 256
__global__ __launch_bounds__(64) void gprep_centroids_kernel(const float* __restrict__ B, const float* __restrict__ mu,
                                                            const float* __restrict__ scale, u32x4* __restrict__ c1,
                                                            u32x4* __restrict__ cnorm, unsigned* __restrict__ cmax2_bits,
                                                            int* __restrict__ cflag, int d, int n, int KAp, int euclid) {
  const int unit = blockIdx.x, lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  const int c = unit * 32 + l31;
  const float s = scale[0];
  float N = 0.f, sraw = 0.f;
  if (c < n) {  // |c'|^2 in double, rounded once: a chain of d fp32 roundings would be a term of the bound
    double Nd = 0.0;
    for (int k = 0; k < d; ++k) {
      const float y = B[(int64_t)k * n + c];
      const float cc = (y - mu[k]) * s;
      Nd += (double)cc * (double)cc;
      sraw = fmaf(y, y, sraw);
    }
    N = (float)Nd;
  }
  float c2m = 0.f;  // |C - Ch|^2: what the fast path drops of this centroid
  int bad = 0;
  {
    bf16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) {
      __bf16 p1, p2, p3;
      split3_bf16(c < n ? (euclid ? -N : 0.f) : -3.0e38f, p1, p2, p3);  // inner product: no norm
      f[0] = p1;
      f[1] = p2;
      f[2] = p3;
    }
    cnorm[(int64_t)unit * 64 + lane] = __builtin_bit_cast(u32x4, f);
  }
  if (c < n) {
    bad |= !(N <= 3.0e38f) | !(sraw <= 3.0e38f);
    if (half == 0 && !bad) {
      atomicMax(cmax2_bits, __float_as_uint(N));
      atomicMax(cmax2_bits + 1, __float_as_uint(sraw));
    }
  }
  for (int st = 0; st < KAp; ++st) {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * st + 8 * half + j;
      const float C = (k < d && c < n) ? (euclid ? 2.f : 1.f) * ((B[(int64_t)k * n + c] - mu[k]) * s) : 0.f;
      bad |= !(fabsf(C) <= 65000.f);
      const _Float16 hh = (_Float16)C;
      const float r = C - (float)hh;
      h[j] = hh;
      c2m = fmaf(r, r, c2m);
    }
    c1[((int64_t)unit * KAp + st) * 64 + lane] = __builtin_bit_cast(u32x4, h);
  }
  c2m += __shfl_xor(c2m, 32, 64);
  if (half == 0 && c < n && !bad) atomicMax(cmax2_bits + 2, __float_as_uint(c2m));
  if (bad) atomicOr(cflag, 1);
}

// pass 2's point operand: the listed points' hi fragments, compact: position pos of the list -> tile
// pos / 32, row pos % 32.  grid (ceil(cap / 32)), one wave per compact tile
__global__ __launch_bounds__(64) void ggather_kernel(const u32x4* __restrict__ hi, const int* __restrict__ list,
                                                    const int* __restrict__ count, u32x4* __restrict__ out, int KAp,
                                                    int cap) {
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  int cnt = *count;
  cnt = cnt < cap ? cnt : cap;
  const int64_t pos = (int64_t)blockIdx.x * 32 + l31;
  if ((int64_t)blockIdx.x * 32 >= cnt) return;
  const int p = pos < cnt ? list[pos] : -1;
  const int64_t src = p >= 0 ? ((int64_t)(p >> 5) * KAp) * 64 + half * 32 + (p & 31) : 0;
  const u32x4 z = {0u, 0u, 0u, 0u};
  u32x4* o = out + ((int64_t)blockIdx.x * KAp) * 64 + lane;
  for (int st = 0; st < KAp; ++st) o[(int64_t)st * 64] = p >= 0 ? hi[src + (int64_t)st * 64] : z;
}

// fold the partial results of a point and decide; an undecided point is listed with its candidate
// threshold.  grid (ceil(m / 256))
struct GDecideArgs {
  const float2* part_b;
  const int* part_i;
  int n_part;
  int64_t rows;                // stride of the partial tables
  const float4* norms;         // [m]: |a'|^2, |x|^2, |a' - ah|^2
  const unsigned* cmax2_bits;  // max N, max |c|^2, max |C - Ch|^2
  const float* scale;
  const int* flag;
  const int* cflag;
  int64_t* inds;
  float* vals;
  int* list;
  int* count;
  float* thr;  // [cap]
  int cap;
  int m;
  float eps, eps_exact, eta;  // eps: everything but the dropped pieces, relative to (|a'| + |c'|max)^2
  int euclid;                 // 0: inner product (no centring, C = c', no norms: the bounds hold a fortiori)
};
// The bound (the derivation of the header comment, with the refinements that matter at d ~ 1000):
//  * the dropped products are bounded by what was actually dropped -- |a' - ah| of the point (split kernel)
//    and max |C - Ch| over the centroids (prep kernel):
//        |sum (a C - ah Ch)| <= |a' - ah| (|Ch|max + |C - Ch|max) + |a'| |C - Ch|max,   |Ch| <= (1 + 2^-11) 2 |c'|
//    -- about a third of the worst case 2^-11 (|a'| + |c'|max)^2;
//  * accumulation: an MFMA adds 16 products and the accumulator, <= 17 roundings of <= 2^-23 (truncation
//    allowed for) of the running magnitude, which is <= sum |a_k C_k| + N <= |a'| 2 |c'| + |c'|^2: at most
//    (|a'| + |c'|)^2 / 2 for the products: (17 (KS + 1) + 8) 2^-24;
//  * N = fl(|c'|^2) is summed in double and rounded once (2^-24 N);
//  * pass 1 compares KEYS (2^-17 |v| off, |v| <= (.)^2), pass 2 the values themselves against the key of
//    the best minus 2 delta: the exact winner's value is within delta of its exact value, the best key within
//    delta of ITS exact value, so the winner is at or above the threshold (lowered by one more key error
//    and a denormal: delta = 0 -- all-zero data -- must still list the best itself).
__global__ __launch_bounds__(256) void gdecide_kernel(GDecideArgs a) {
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = pos < a.m;
  const int p = valid ? (int)pos : 0;
  float B1 = -INFINITY, B2 = -INFINITY;
  int idx = 0;
  if (valid)
    for (int c = 0; c < a.n_part; ++c) {
      const float2 v = a.part_b[(int64_t)c * a.rows + pos];
      const int i = a.part_i[(int64_t)c * a.rows + pos];
      const float n2 = fmaxf(fminf(B1, v.x), fmaxf(B2, v.y));
      idx = (v.x > B1 || (v.x == B1 && i < idx)) ? i : idx;
      B1 = fmaxf(B1, v.x);
      B2 = n2;
    }
  const float s = a.scale[0];
  const float cn = sqrtf(__uint_as_float(a.cmax2_bits[0])), cnr = sqrtf(__uint_as_float(a.cmax2_bits[1]));
  const float4 n2 = a.norms[p];
  const float an = sqrtf(n2.x), anr = sqrtf(n2.y) * s;
  const float t1 = an + cn, t2 = anr + cnr * s;
  const float a2 = sqrtf(n2.z), c2 = sqrtf(__uint_as_float(a.cmax2_bits[2]));
  const float cscale = a.euclid ? 2.002f : 1.001f;
  const float dropped = a2 * (cscale * cn + c2) + 1.001f * an * c2;
  float delta = 1.25f * (dropped + a.eps * t1 * t1 + a.eta * (2.f * cn + an) + a.eps_exact * t2 * t2);
  if ((a.flag[0] | a.cflag[0]) != 0) delta = INFINITY;
  if (valid) {
    a.inds[p] = idx;
    if (a.vals) a.vals[p] = (a.euclid ? B1 - n2.x : B1) * ((1.f / s) * (1.f / s));
  }
  const bool listed = valid && !(B1 - B2 > 2.f * delta);
  const unsigned long long mk = __ballot(listed);
  if (mk) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mk) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(a.count, __popcll(mk));
    base = __shfl(base, leader, 64);
    if (listed) {
      const int slot = base + __popcll(mk & ((1ull << lane) - 1ull));
      a.list[slot] = p;
      // (a flagged problem -- delta = inf, keys possibly inf / NaN -- emits no candidates: gdecode_kernel sends
      // its whole list to the exact kernel)
      if (slot < a.cap)
        a.thr[slot] = (a.flag[0] | a.cflag[0]) != 0 ? __builtin_nanf("")  // (no value compares >= NaN, not even inf)
                                                    : B1 - 2.f * delta - (fabsf(B1) * (1.0f / 65536.0f) + 1.0e-30f);
    }
  }
}

// pass 3 reads ROWS: a (point, centroid) pair needs one column of A [d][m] and one of B [d][n] -- 4 useful bytes
// per 64-byte sector as they lie (7 ms for 400 000 pairs at d = 960).  rows_kernel copies the columns it is given
// into rows out[j][dp] (dp = d rounded up to 16, zero padded): all centroids once per call (63 MB at 16 384 x
// 960), and the listed points (their columns are ascending and ~5 apart: a few sectors per dimension).
// grid (ceil(count / 32)), 64 lanes: lane (column i, half) gathers 16 consecutive dimensions, stores 64 B.
__global__ __launch_bounds__(64) void rows_kernel(const float* __restrict__ M, int64_t cols, const int* __restrict__ list,
                                                 const int* __restrict__ count, int cap, float* __restrict__ out, int d,
                                                 int dp) {
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  int cnt = cap;
  if (count) {
    cnt = *count;
    cnt = cnt < cap ? cnt : cap;
  }
  if ((int64_t)blockIdx.x * 32 >= cnt) return;
  const int j = blockIdx.x * 32 + l31;
  const bool valid = j < cnt;
  const int64_t col = valid ? (list ? list[j] : j) : 0;
  const float* src = M + col;
  float* dst = out + (int64_t)j * dp;
  for (int k0 = 16 * half; k0 < dp; k0 += 32) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (k0 + u < d) ? src[(int64_t)(k0 + u) * cols] : 0.f;
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(dst + k0 + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
  }
}

// pass 3: the exact kernel's value of every (listed point, candidate centroid) pair, folded into keys[point].
// A wave takes 32 pairs: lane (pair i, half); half 0 streams the point's row, half 1 the centroid's row -- each
// the ascending-k fma chain of its squared norm as max_sim_kernel forms it -- and per pair of dimensions
// the halves swap what the other needs as MFMA operand (row i = centroid of pair i, column i = point of pair
// i: the diagonal of the 32 x 32 tile holds the 32 results).  The zero padding beyond d adds +0 to chains and
// products alike.
__global__ __launch_bounds__(256) void pair_exact_kernel(const float* __restrict__ Xt, const float* __restrict__ Bt,
                                                        const uint2* __restrict__ pairs, const int* __restrict__ pair_count,
                                                        int pair_cap, const int* __restrict__ list,
                                                        unsigned long long* __restrict__ keys, int dp, int euclid) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  int cnt = *pair_count;
  cnt = cnt < pair_cap ? cnt : pair_cap;
  const int tiles = (cnt + 31) >> 5;
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int q = tile * 32 + l31;
    const bool valid = q < cnt;
    const uint2 pr = valid ? pairs[q] : make_uint2(0u, 0u);
    const int p = valid ? list[pr.x] : 0;
    const int c = (int)pr.y;
    const float4* src = reinterpret_cast<const float4*>(half ? Bt + (int64_t)c * dp : Xt + (int64_t)pr.x * dp);
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float chain = 0.f;  // half 0: |x|^2, half 1: |c|^2
    for (int k0 = 0; k0 < dp; k0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 t = src[(k0 >> 2) + u];
        v[4 * u] = t.x;
        v[4 * u + 1] = t.y;
        v[4 * u + 2] = t.z;
        v[4 * u + 3] = t.w;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) chain = fmaf(v[u], v[u], chain);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // lane (i, half) multiplies dimension 2 j + half: half 0 keeps x[2j] and sends x[2j+1], half 1
        // keeps c[2j+1] and sends c[2j]
        const float own = half ? v[2 * j + 1] : v[2 * j];
        const float got = __shfl_xor(half ? v[2 * j] : v[2 * j + 1], 32, 64);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? own : got, half ? got : own, acc, 0, 0, 0);
      }
    }
    const float other = __shfl_xor(chain, 32, 64);
    const float x2 = half ? other : chain, c2 = half ? chain : other;
    // D[row i][col i]: column = l31, row = (r & 3) + 8 (r >> 2) + 4 half
    const int rsel = (l31 & 3) | ((l31 >> 3) << 2);
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dot = (r == rsel) ? acc[r] : dot;
    float v = dot;
    if (euclid) {
      v = 2.f * v;
      v = v - x2;
      v = v - c2;
    }
    v = v + 0.f;  // (-0 -> +0: the key orders by bits)
    if (valid && half == ((l31 >> 2) & 1)) {
      const unsigned fb = __float_as_uint(v);
      const unsigned ordered = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);
      atomicMax(keys + p, ((unsigned long long)ordered << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)c));
    }
  }
}

// labels / values of the listed points from their keys; then the fallback switch: the exact kernel runs
// over the level-1 list iff pairs were dropped or the list outgrew the compact array.  grid (ceil(m / 256))
__global__ __launch_bounds__(256) void gdecode_kernel(const int* __restrict__ list, const int* __restrict__ count,
                                                     const unsigned long long* __restrict__ keys,
                                                     float* __restrict__ vals, int64_t* __restrict__ inds, int m, int cap,
                                                     const int* __restrict__ overflow, const int* __restrict__ flag,
                                                     const int* __restrict__ cflag, int* __restrict__ count_fb) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int cnt = *count < m ? *count : m;
  if (p == 0) *count_fb = (*overflow != 0 || cnt > cap || (flag[0] | cflag[0]) != 0) ? cnt : 0;
  if (p >= cnt) return;
  const int i = list[p];
  const unsigned long long key = keys[i];
  if (key == 0ull) return;  // (beyond the compact array: the fallback's)
  const unsigned ordered = (unsigned)(key >> 32);
  const unsigned fb = (ordered & 0x80000000u) ? (ordered & 0x7FFFFFFFu) : ~ordered;
  inds[i] = (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  if (vals) vals[i] = __uint_as_float(fb);
}

struct WideLayout {
  int KS, KAp, U, ncb, ysplit, cap2, cap3, pair_cap;
  int64_t T;
  size_t mu_off, scale_off, flag_off, maxbits_off, cmax_off, cflag_off, count1_off, countfb_off, npairs_off, oflag_off,
      phi_off, norms_off, c1_off, cnorm_off, p2_off, thr_off, pairs_off, partb_off, parti_off, list1_off, keys_off, ac_off,
      bt_off, xt_off, total;
  int dp;
};
static WideLayout wide_layout(int d, int64_t m, int n) {
  WideLayout L;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  L.KS = (d + 15) / 16;
  L.KAp = (L.KS + kGK - 1) / kGK * kGK;
  L.ncb = (n + 255) / 256;
  L.U = L.ncb * 8;
  L.T = (m + 255) / 256 * 8;  // whole blocks of 256 points
  L.ysplit = L.ncb >= 8 ? 8 : (L.ncb >= 4 ? 4 : (L.ncb >= 2 ? 2 : 1));  // (gemm_kernel's block mapping)
  if (const char* e = TPQ_AB_ENV("TPQ_WIDE_YSPLIT")) L.ysplit = atoi(e);  // (A/B; 1, 2, 4, 8)
  L.cap2 = (int)(((m / 2 > 8192 ? m / 2 : 8192) + 255) / 256 * 256);     // pass 2's compact array, in points
  if (L.cap2 > L.T * 32) L.cap2 = (int)(L.T * 32);
  L.pair_cap = 4 * L.cap2 > 65536 ? 4 * L.cap2 : 65536;
  L.cap3 = (int)((m / 16 + 127) / 128 * 128);  // the fallback's compact copy
  if (L.cap3 < 128) L.cap3 = 128;
  L.mu_off = 0;
  L.scale_off = up((size_t)(d > 128 ? d : 128) * 4);
  L.flag_off = L.scale_off + 4;
  L.maxbits_off = L.flag_off + 4;
  L.cmax_off = L.maxbits_off + 4;
  L.cflag_off = L.cmax_off + 12;
  L.count1_off = L.cflag_off + 4;
  L.countfb_off = L.count1_off + 4;
  L.npairs_off = L.countfb_off + 4;
  L.oflag_off = L.npairs_off + 4;
  L.phi_off = up(L.oflag_off + 4);
  L.norms_off = up(L.phi_off + (size_t)L.T * L.KAp * 1024);
  L.c1_off = up(L.norms_off + (size_t)L.T * 32 * 16);
  L.cnorm_off = up(L.c1_off + (size_t)L.U * L.KAp * 1024);
  L.p2_off = up(L.cnorm_off + (size_t)L.U * 1024);
  L.thr_off = up(L.p2_off + (size_t)(L.cap2 / 32) * L.KAp * 1024);
  L.pairs_off = up(L.thr_off + (size_t)L.cap2 * 4);
  L.partb_off = up(L.pairs_off + (size_t)L.pair_cap * 8);
  L.parti_off = up(L.partb_off + (size_t)2 * L.ysplit * (size_t)(L.T * 32) * 8);
  L.list1_off = up(L.parti_off + (size_t)2 * L.ysplit * (size_t)(L.T * 32) * 4);
  L.keys_off = up(L.list1_off + (size_t)m * 4);
  L.ac_off = up(L.keys_off + (size_t)m * 8);
  L.dp = (d + 15) / 16 * 16;
  L.bt_off = up(L.ac_off + (size_t)d * L.cap3 * 4);            // [n][dp] f32: the centroids as rows
  L.xt_off = up(L.bt_off + (size_t)n * L.dp * 4);              // [cap2][dp] f32: the listed points as rows
  L.total = up(L.xt_off + (size_t)L.cap2 * L.dp * 4);
  return L;
}

// blocks of gemm_kernel: whole groups of 8 XCDs x 32 blocks (P point blocks x ysplit ranges each)
static unsigned gemm_grid(int pblocks, int ysplit) {
  const int per_group = 8 * (32 / ysplit);
  return (unsigned)((pblocks + per_group - 1) / per_group) * 256u;
}

static int run_wide(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n, int euclid,
                    char* ws, const WideLayout& L, hipStream_t st) {
  float* mu = reinterpret_cast<float*>(ws + L.mu_off);
  float* scale = reinterpret_cast<float*>(ws + L.scale_off);
  int* flag = reinterpret_cast<int*>(ws + L.flag_off);
  unsigned* maxbits = reinterpret_cast<unsigned*>(ws + L.maxbits_off);
  unsigned* cmax = reinterpret_cast<unsigned*>(ws + L.cmax_off);
  int* cflag = reinterpret_cast<int*>(ws + L.cflag_off);
  int* count1 = reinterpret_cast<int*>(ws + L.count1_off);
  int* count_fb = reinterpret_cast<int*>(ws + L.countfb_off);
  int* n_pairs = reinterpret_cast<int*>(ws + L.npairs_off);
  int* oflag = reinterpret_cast<int*>(ws + L.oflag_off);
  u32x4* phi = reinterpret_cast<u32x4*>(ws + L.phi_off);
  float4* norms = reinterpret_cast<float4*>(ws + L.norms_off);
  u32x4* c1 = reinterpret_cast<u32x4*>(ws + L.c1_off);
  u32x4* cnorm = reinterpret_cast<u32x4*>(ws + L.cnorm_off);
  u32x4* p2 = reinterpret_cast<u32x4*>(ws + L.p2_off);
  float* thr = reinterpret_cast<float*>(ws + L.thr_off);
  uint2* pairs = reinterpret_cast<uint2*>(ws + L.pairs_off);
  float2* part_b = reinterpret_cast<float2*>(ws + L.partb_off);
  int* part_i = reinterpret_cast<int*>(ws + L.parti_off);
  int* list1 = reinterpret_cast<int*>(ws + L.list1_off);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + L.keys_off);
  float* Ac = reinterpret_cast<float*>(ws + L.ac_off);
  float* Bt = reinterpret_cast<float*>(ws + L.bt_off);
  float* Xt = reinterpret_cast<float*>(ws + L.xt_off);
  int rc = check_hip(hipMemsetAsync(ws, 0, L.phi_off, st), "coarse_assign (wide) memset");
  if (rc) return rc;
  rc = check_hip(hipMemsetAsync(keys, 0, (size_t)m * 8, st), "coarse_assign (wide) keys memset");
  if (rc) return rc;
  // (rows beyond m / beyond the list in the last block of 256: an MFMA column depends on its own point only,
  // and those columns are never written out)
  if (euclid) {  // (inner products are not shift invariant: mu stays 0)
    hipLaunchKernelGGL(mu_kernel, dim3(d, 1), dim3(256), 0, st, B, mu, d, n);
    TPQ_LAUNCH_CHECK("lloyd mu_kernel");
  }
  int chunks = (int)(8192 / (int64_t)d);
  if (chunks < 1) chunks = 1;
  if ((int64_t)chunks * 4096 > m) chunks = (int)((m + 4095) / 4096);
  hipLaunchKernelGGL(maxabs_kernel, dim3(chunks, d, 1), dim3(256), 0, st, A, mu, maxbits, flag, d, m);
  TPQ_LAUNCH_CHECK("lloyd maxabs_kernel");
  hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(64), 0, st, maxbits, flag, scale, 1);
  TPQ_LAUNCH_CHECK("lloyd scale_kernel");
  hipLaunchKernelGGL(gsplit_points_kernel, dim3((unsigned)((L.T + 3) / 4)), dim3(256), 0, st, A, mu, scale, phi, norms,
                     d, m, L.T, L.KAp);
  TPQ_LAUNCH_CHECK("lloyd gsplit_points_kernel");
  hipLaunchKernelGGL(gprep_centroids_kernel, dim3(L.U), dim3(64), 0, st, B, mu, scale, c1, cnorm, cmax, cflag, d, n,
                     L.KAp, euclid);
  TPQ_LAUNCH_CHECK("lloyd gprep_centroids_kernel");
  const size_t lds = (size_t)2 * kGStage + 8 * 1024 + sizeof(PairList);
  static const int tile_ct = TPQ_AB_ENV("TPQ_WIDE_CT") ? atoi(TPQ_AB_ENV("TPQ_WIDE_CT")) : 2;  // (A/B of the wave tile)
  const auto k_top2 = tile_ct == 4 ? gemm_kernel<false, 4> : gemm_kernel<false, 2>;
  const auto k_cand = tile_ct == 4 ? gemm_kernel<true, 4> : gemm_kernel<true, 2>;
  const int gemm_threads = tile_ct == 4 ? 256 : 512;
  rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(k_top2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds), "lloyd gemm_kernel attr");
  if (rc) return rc;
  rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cand), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds), "lloyd gemm_kernel attr");
  if (rc) return rc;
  const int64_t rows1 = L.T * 32;
  // (gdecide_kernel's comment) accumulation, N, shift rounding, 6-bit keys
  const float eps = 1.001f * (float)(17 * (L.KS + 1) + 8) / 16777216.0f + 1.0f / 8388608.0f + 1.0f / 4194304.0f +
                    1.0f / 131072.0f;
  {  // pass 1
    const int pblocks = (int)(L.T / 8);
    GemmArgs ga{c1, phi, cnorm, part_b, part_i, L.KAp, L.ncb, L.ysplit, pblocks, rows1, nullptr, nullptr, nullptr, nullptr,
                0, nullptr, n};
    hipLaunchKernelGGL(k_top2, dim3(gemm_grid(pblocks, L.ysplit)), dim3(gemm_threads), lds, st, ga);
    TPQ_LAUNCH_CHECK("lloyd gemm_kernel");
    // (ranges of ceil(ncb / ysplit) centroid blocks: the last ones may be empty and write nothing)
    const int cb_per = (L.ncb + L.ysplit - 1) / L.ysplit, yused = (L.ncb + cb_per - 1) / cb_per;
    GDecideArgs da{part_b, part_i, 2 * yused, rows1, norms, cmax, scale, flag, cflag, inds, vals, list1, count1,
                   thr, L.cap2, (int)m, eps, (float)(d + 4) / 16777216.0f, sqrtf((float)d) / 8192.0f, euclid};
    hipLaunchKernelGGL(gdecide_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, da);
    TPQ_LAUNCH_CHECK("lloyd gdecide_kernel");
  }
  {  // pass 2: the candidates of the undecided points
    hipLaunchKernelGGL(ggather_kernel, dim3((unsigned)(L.cap2 / 32)), dim3(64), 0, st, phi, list1, count1, p2, L.KAp,
                       L.cap2);
    TPQ_LAUNCH_CHECK("lloyd ggather_kernel");
    GemmArgs ga{c1, p2, cnorm, nullptr, nullptr, L.KAp, L.ncb, L.ysplit, L.cap2 / 256, (int64_t)L.cap2, count1, thr, pairs,
                n_pairs, L.pair_cap, oflag, n};
    hipLaunchKernelGGL(k_cand, dim3(gemm_grid(L.cap2 / 256, L.ysplit)), dim3(gemm_threads), lds, st, ga);
    TPQ_LAUNCH_CHECK("lloyd gemm_kernel (candidates)");
  }
  // pass 3: exact values of the pairs
  hipLaunchKernelGGL(rows_kernel, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, st, B, (int64_t)n,
                     static_cast<const int*>(nullptr), static_cast<const int*>(nullptr), n, Bt, d, L.dp);
  TPQ_LAUNCH_CHECK("lloyd rows_kernel");
  hipLaunchKernelGGL(rows_kernel, dim3((unsigned)(L.cap2 / 32)), dim3(64), 0, st, A, m, list1, count1, L.cap2, Xt, d, L.dp);
  TPQ_LAUNCH_CHECK("lloyd rows_kernel");
  hipLaunchKernelGGL(pair_exact_kernel, dim3(2048), dim3(256), 0, st, Xt, Bt, pairs, n_pairs, L.pair_cap, list1, keys,
                     L.dp, euclid);
  TPQ_LAUNCH_CHECK("lloyd pair_exact_kernel");
  hipLaunchKernelGGL(gdecode_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, list1, count1, keys, vals, inds,
                     (int)m, L.cap2, oflag, flag, cflag, count_fb);
  TPQ_LAUNCH_CHECK("lloyd gdecode_kernel");
  // (normally over zero points)
  return launch_max_sim_list(A, B, vals, inds, 1, d, (int)m, n, euclid, list1, count_fb, keys, Ac, L.cap3, st);
}


// ---- the candidate route of the narrow path (d <= 128, many centroids) ---------------------------------------
// Level 1 of a chunked problem leaves 3-10 % of the points undecided.  Levels 2 and 3 (refine_stream_kernel:
// three products against every chunk; then the exact kernel over ALL centroids for what is left) cost 0.8 +
// 1.8 of the 7.4 ms at 1 M x 16 384 x 128.  Instead: the listed points' hi pieces stay in registers (64 points
// per wave: two column tiles share every centroid fragment), the hi fragments and -N of all chunks stream
// through a double-buffered LDS ring, and every value at or above the point's threshold (level 1's best key
// minus twice ITS bound, decide_kernel) is a candidate pair for pair_exact_kernel.  One product, no top-2.
struct CandStreamArgs {
  const u32x4* hi;       // [T][Q][32][64 B] (the points' hi pieces, prep layout)
  const u32x4* frags;    // [units][2 KS + 1][64]
  const int* list_in;
  const int* count_in;
  const float* thr;      // [cap]
  int cap;               // listed positions handled here: < min(count, cap)
  int n_half;            // half chunks (4 units each)
  int n;                 // centroids (units beyond are padding)
  int64_t T;
};
constexpr int kCandPoints = kWaves * 64;  // listed positions per block

template <int KS>
__global__ __launch_bounds__(kWaves * 64) void cand_stream_kernel(CandStreamArgs c, GemmArgs ga) {
  constexpr int FPU = 2 * KS + 1, FL = KS + 1, HB = 4 * FL * 1024, Q = (KS + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 half chunks (-N + hi fragments), PairList
  int cnt = *c.count_in;
  cnt = cnt < c.cap ? cnt : c.cap;
  if ((int64_t)blockIdx.x * kCandPoints >= cnt) return;  // block-uniform
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  PairList* pl = reinterpret_cast<PairList*>(smem + 2 * HB);
  if (threadIdx.x == 0) pl->n = 0;
  auto stage = [&](int h) {  // half chunk h -> buffer h & 1: per unit -N, then the hi piece of every k-step
    const char* src = reinterpret_cast<const char*>(c.frags) + (size_t)h * 4 * FPU * 1024;
    char* dst = smem + (h & 1) * HB;
    for (int f = wave; f < 4 * FL; f += kWaves) {
      const int unit = f / FL, j = f % FL;
      const int sf = unit * FPU + (j ? 2 * j - 1 : 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + sf * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
    }
  };
  stage(0);
  const int64_t slice = c.T * Q * 2048;
  const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(c.hi)), 0, (int)slice, 0x00020000);
  f16x8 xs[KS][2];
  float thr[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int64_t pos = (int64_t)blockIdx.x * kCandPoints + wave * 64 + ct * 32 + l31;
    const int p = pos < cnt ? c.list_in[pos] : -1;
    thr[ct] = p >= 0 ? c.thr[pos] : __builtin_nanf("");  // (NaN: no value compares >= it, not even an inf)
    const int voff = p >= 0 ? (p >> 5) * (Q * 2048) + (p & 31) * 64 + half * 16 : 0x7ffffff0;
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      xs[st][ct] = __builtin_bit_cast(
          f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_hi, voff, (st >> 1) * 2048 + (st & 1) * 32, 0));
    });
  }
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int h = 0; h < c.n_half; ++h) {
    __syncthreads();  // half chunk h has landed (vmcnt(0) + barrier); everyone is done with the other buffer
    if (h + 1 < c.n_half) stage(h + 1);
    // (pairs are appended in every stage, so pl->n is not stable anywhere: the decision to flush is an OR over
    // the block, taken every eighth stage -- ~70 appends per block in between against 2 048 spare entries)
    if ((h & 7) == 7 && __syncthreads_or(pl->n >= kPairList / 2)) flush_pairs(ga, pl, blockIdx.x * (unsigned)kCandPoints);
    const u32x4* base = reinterpret_cast<const u32x4*>(smem + (h & 1) * HB) + lane;
#pragma unroll
    for (int U = 0; U < 4; ++U) {
      const u32x4* up = base + U * FL * 64;
      f32x16 acc[2] = {zero, zero};
      static_for<0, KS>([&](auto s_c) {
        constexpr int st = decltype(s_c)::value;
        const f16x8 cf = __builtin_bit_cast(f16x8, up[(1 + st) * 64]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cf, xs[st][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cf, xs[st][1], acc[1], 0, 0, 0);
      });
      const bf16x8 nf = __builtin_bit_cast(bf16x8, up[0]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nf, bones, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nf, bones, acc[1], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float mx = acc[ct][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[ct][r]);
        if (__ballot(mx >= thr[ct]) == 0ull) continue;  // (nearly every tile)
        const unsigned rowbits = (unsigned)(wave * 64 + ct * 32 + l31) << 22;
        const int cbase = (4 * h + U) * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cen = cbase + (r & 3) + 8 * (r >> 2);
          const bool hit = acc[ct][r] >= thr[ct] && cen < c.n;
          const unsigned long long mk = __ballot(hit);
          if (mk) {
            const int leader = __ffsll((long long)mk) - 1;
            int b0 = 0;
            if (lane == leader) b0 = atomicAdd(&pl->n, __popcll(mk));  // LDS
            b0 = __shfl(b0, leader, 64);
            const int slot = b0 + __popcll(mk & ((1ull << lane) - 1ull));
            if (hit && slot < kPairList) pl->item[slot] = rowbits | (unsigned)cen;
          }
        }
      }
    }
  }
  __syncthreads();
  flush_pairs(ga, pl, blockIdx.x * (unsigned)kCandPoints);
}

template <int KS>
static int run_cand_tail(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n, char* ws,
                         const AssignLayout& L, const u32x4* hi, const u32x4* frags, hipStream_t st) {
  int* count1 = reinterpret_cast<int*>(ws + L.count1_off);
  int* n_pairs = reinterpret_cast<int*>(ws + L.npairs_off);
  int* oflag = reinterpret_cast<int*>(ws + L.oflag_off);
  int* count_fb = reinterpret_cast<int*>(ws + L.countfb_off);
  int* list1 = reinterpret_cast<int*>(ws + L.list1_off);
  float* thr = reinterpret_cast<float*>(ws + L.thr_off);
  uint2* pairs = reinterpret_cast<uint2*>(ws + L.pairs_off);
  float* Bt = reinterpret_cast<float*>(ws + L.bt_off);
  float* Xt = reinterpret_cast<float*>(ws + L.xt_off);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + L.keys_off);
  float* Ac = reinterpret_cast<float*>(ws + L.ac_off);
  const size_t lds = (size_t)2 * 4 * (KS + 1) * 1024 + sizeof(PairList);
  auto kernel = cand_stream_kernel<KS>;
  int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds), "lloyd cand_stream_kernel attr");
  if (rc) return rc;
  CandStreamArgs ca{hi, frags, list1, count1, thr, L.cap2, 2 * L.chunks, n, L.P.T};
  GemmArgs ga{};
  ga.pairs = pairs;
  ga.pair_count = n_pairs;
  ga.pair_cap = L.pair_cap;
  ga.overflow = oflag;
  hipLaunchKernelGGL(kernel, dim3((unsigned)(L.cap2 / kCandPoints)), dim3(kWaves * 64), lds, st, ca, ga);
  TPQ_LAUNCH_CHECK("lloyd cand_stream_kernel");
  hipLaunchKernelGGL(rows_kernel, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, st, B, (int64_t)n,
                     static_cast<const int*>(nullptr), static_cast<const int*>(nullptr), n, Bt, d, L.dp);
  TPQ_LAUNCH_CHECK("lloyd rows_kernel");
  hipLaunchKernelGGL(rows_kernel, dim3((unsigned)(L.cap2 / 32)), dim3(64), 0, st, A, m, list1, count1, L.cap2, Xt, d, L.dp);
  TPQ_LAUNCH_CHECK("lloyd rows_kernel");
  hipLaunchKernelGGL(pair_exact_kernel, dim3(2048), dim3(256), 0, st, Xt, Bt, pairs, n_pairs, L.pair_cap, list1, keys,
                     L.dp, 1);
  TPQ_LAUNCH_CHECK("lloyd pair_exact_kernel");
  hipLaunchKernelGGL(gdecode_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, list1, count1, keys, vals, inds,
                     (int)m, L.cap2, oflag, reinterpret_cast<const int*>(ws + L.prep_off + L.P.flag_off),
                     reinterpret_cast<const int*>(ws + L.cflag_off), count_fb);
  TPQ_LAUNCH_CHECK("lloyd gdecode_kernel");
  // (normally over zero points: pair lists that overflowed, more listed points than the row copies hold, or a
  // flagged problem)
  return launch_max_sim_list(A, B, vals, inds, 1, d, (int)m, n, 1, list1, count_fb, keys, Ac, L.cap, st);
}

// ---- the coarse step of search(): fast similarities of every (query, cell) pair (probe_fast.h) ---------------
// coarse_kernel's loop -- hi pieces only, one product per k-step, the -N MFMA, two column tiles per A operand,
// the chunk's fragments staged once per block by LDS-DMA -- with another epilogue: instead of the top-2 update (2.5
// VALU instructions per value, what bounds level 1) the 16 values a lane holds of its query are stored as four
// 16-byte pieces of the query's row (rows 8 g + 4 half + j of a 32 x 32 tile are four consecutive cells), and the
// maximum over each 128-cell group is kept for the row select's group filter.  grid (query blocks, 256-cell chunks);
// a block walks n_wide wide tiles per wave (small query batches: one, so that 10 000 queries x 64 chunks are 1 280 blocks).
// (see the epilogue of probe_sims_kernel)
#define TPQ_STORE_PAD() asm volatile("s_nop 7" ::: "memory")

struct ProbeSimsArgs {
  const u32x4* hi;
  const u32x4* frags;
  _Float16* sims;        // [nq][n_cells] f' x qscale[q], fp16
  const float* qscale;   // [nq] the power of two that puts |f'| <= (|a'| + |c'|max)^2 of the query below 2^15
  float* gmax;           // [nq][n_groups] maxima of the UNROUNDED f' (fp32, unscaled)
  int nq, n_cells, n_groups, n_wide;
  int64_t T;
  int chunk_frag_stride;
};

template <int KS, int GSH>   // GSH: log2 of the cells per group of the maxima (5: one unit, 7: four)
__global__ __launch_bounds__(kWaves * 64, 2) void probe_sims_kernel(ProbeSimsArgs a) {
  constexpr int NG = 256 >> GSH;   // groups per 256-cell chunk
  constexpr int FPU = 2 * KS + 1;  // fragments per unit in global memory
  constexpr int FL = KS + 1;       // ... in LDS
  constexpr int Q = (KS + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int chunk = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  {
    const char* src = reinterpret_cast<const char*>(a.frags) + (size_t)chunk * a.chunk_frag_stride * 16;
    for (int f = wave; f < 8 * FL; f += kWaves) {
      const int unit = f / FL, j = f % FL;
      const int sf = unit * FPU + (j ? 2 * j - 1 : 0);  // -N, then the hi piece of k-step j - 1
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + sf * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + f * 1024), 16, 0, 0);
    }
  }
  const int64_t slice = a.T * Q * 2048;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.hi)), 0, (int)slice, 0x00020000);
  const int n_wide = a.n_wide;
  auto wide_of = [&](int t) -> int64_t { return ((int64_t)blockIdx.x * n_wide + t) * kWaves + wave; };
  auto frag_voff = [&](int t) -> int {
    const int64_t wt = wide_of(t);
    return (t < n_wide && 2 * wt < a.T) ? (int)(2 * wt * Q * 2048) + l31 * 64 + half * 16 : 0x7ffffff0;
  };
  f16x8 xsb[2][2][KS];  // [buffer][column tile][k-step]
  auto load_frag = [&](int voff, auto e_c, f16x8 (&dst)[2][KS]) {
    constexpr int e = decltype(e_c)::value, ct = e / KS, st = e % KS;
    dst[ct][st] = __builtin_bit_cast(
        f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ct * Q * 2048 + (st >> 1) * 2048 + (st & 1) * 32, 0));
  };
  {
    const int voff = frag_voff(0);
    static_for<0, 2 * KS>([&](auto e_c) { load_frag(voff, e_c, xsb[0]); });
  }
  __syncthreads();  // fragments (vmcnt(0) of the DMA) are in LDS
  const u32x4* fp = reinterpret_cast<const u32x4*>(smem) + lane;
  auto ldsf = [&](const u32x4* p) -> f16x8 { return __builtin_bit_cast(f16x8, *p); };
  f32x16 acc[2];
  f16x8 a0 = ldsf(fp + 1 * 64), a1 = a0, aring[3];
  if constexpr (KS > 1) a1 = ldsf(fp + 2 * 64);
  bf16x8 bones = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half == 0) {
    bones[0] = (__bf16)1.0f;
    bones[1] = (__bf16)1.0f;
    bones[2] = (__bf16)1.0f;
  }
  float gm[2][NG];  // [column tile][group of the chunk]
  // the sims rows of this block's queries as ONE buffer resource (base: the block's first query, the chunk's first
  // cell): a lane's stores are buffer_store_dwordx4 at a 32-bit offset -- its row, its half -- plus a compile-time
  // constant; rows beyond nq get an offset beyond the resource's range and are dropped by the hardware (64-bit
  // per-lane pointers and exec-mask predicates put this kernel 319 registers over its budget)
  const int64_t q_block0 = (int64_t)blockIdx.x * n_wide * kWaves * 64;
  const int64_t rows_here = (a.nq - q_block0) < (int64_t)n_wide * kWaves * 64 ? (a.nq - q_block0) : (int64_t)n_wide * kWaves * 64;
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(a.sims + q_block0 * a.n_cells + chunk * 256), 0,
      (int)(rows_here > 0 ? (rows_here - 1) * (int64_t)a.n_cells * 2 + (a.n_cells - chunk * 256) * 2 : 0), 0x00020000);
  float qs[2];     // the lane's query's scale, per column tile
  int svoff[2];    // byte offset of the lane's row (and half) of each column tile inside that resource
  const int units_here = (a.n_cells - chunk * 256 + 31) / 32;  // (n_cells % 32 == 0: whole units)

  auto unit = [&](auto u_c, int voff_next, const f16x8 (&xs)[2][KS], f16x8 (&xsn)[2][KS]) {
    constexpr int U = decltype(u_c)::value;
    const u32x4* up = fp + U * FL * 64;
    const u32x4* upn = fp + ((U + 1) & 7) * FL * 64;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16x8 cfrag = __builtin_bit_cast(bf16x8, up[0]);
    if constexpr (U < 4) {
      constexpr int l0 = (U * 2 * KS) / 4, l1 = ((U + 1) * 2 * KS) / 4;
      static_for<l0, l1>([&](auto e_c) { load_frag(voff_next, e_c, xsn); });
    }
    static_for<0, KS>([&](auto s_c) {
      constexpr int st = decltype(s_c)::value;
      if constexpr (st + 2 < KS) aring[(st + 2) % 3] = ldsf(up + (1 + st + 2) * 64);
      if constexpr (st == 0) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xs[0][0], zero, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xs[1][0], zero, 0, 0, 0);
        a0 = ldsf(upn + 1 * 64);
      } else if constexpr (st == 1) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xs[0][1], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xs[1][1], acc[1], 0, 0, 0);
        a1 = ldsf(upn + 2 * 64);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[st % 3], xs[0][st], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[st % 3], xs[1][st], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfrag, bones, acc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (U < units_here) {  // wave-uniform (the last chunk of a cell count that is not a multiple of 256)
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      u32x4 w[2][2];  // [column tile][pair of pieces]: eight consecutive cells of the lane's query, fp16
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float mx = gm[ct][U >> (GSH - 5)];
        uint32_t pk[4][2];  // the lane's four pieces (cells 8 g + 4 half + 0..3) as fp16 pairs
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]};
          mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
          const f16x2 h0 = {(_Float16)(v[0] * qs[ct]), (_Float16)(v[1] * qs[ct])};
          const f16x2 h1 = {(_Float16)(v[2] * qs[ct]), (_Float16)(v[3] * qs[ct])};
          pk[g][0] = __builtin_bit_cast(uint32_t, h0);
          pk[g][1] = __builtin_bit_cast(uint32_t, h1);
        }
        gm[ct][U >> (GSH - 5)] = mx;
        // lanes l and l + 32 hold the two halves of the same eight cells of the same query: v_permlane32_swap gives the
        // lower lane both halves of piece 2 p and the upper lane both halves of piece 2 p + 1 -- 16-byte stores of eight
        // consecutive cells (as 8-byte stores the kernel is bound by the number of store instructions, not their bytes)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * p][0], pk[2 * p + 1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * p][1], pk[2 * p + 1][1], false, false);
          w[ct][p] = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      }
      // The four stores last and back to back, then TPQ_STORE_PAD: on gfx950 a VALU instruction that overwrites a
      // register of a 16-byte store's data two instructions after the store (all the wait the compiler's hazard rule
      // asks for) reaches the register file before the store has read it -- measured here: with the stores in between
      // the conversions, 9 % of the rows held a later group maximum in the first dword of a piece
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          __builtin_amdgcn_raw_buffer_store_b128(w[ct][p], srsrc, svoff[ct], (U * 32 + 16 * p) * 2, 0);
      TPQ_STORE_PAD();
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  auto tile = [&](int t, auto cb_c) {
    constexpr int CB = decltype(cb_c)::value, NX = 1 - CB;
    const int voff_next = frag_voff(t + 1);
    const int64_t wt = wide_of(t);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int64_t qi = (2 * wt + ct) * 32 + l31;
      svoff[ct] = qi < a.nq ? (int)((qi - q_block0) * a.n_cells * 2) + half * 16 : 0x7ffffff0;
      qs[ct] = a.qscale[qi < a.nq ? qi : 0];
#pragma unroll
      for (int gg = 0; gg < NG; ++gg) gm[ct][gg] = -INFINITY;
    }
    static_for<0, 8>([&](auto u_c) { unit(u_c, voff_next, xsb[CB], xsb[NX]); });
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int64_t qi = (2 * wt + ct) * 32 + l31;
#pragma unroll
      for (int gg = 0; gg < NG; ++gg) {
        const float m2 = fmaxf(gm[ct][gg], __shfl_xor(gm[ct][gg], 32, 64));  // the two halves hold disjoint cells
        const int grp = chunk * NG + gg;
        if (half == 0 && qi < a.nq && grp < a.n_groups) a.gmax[qi * a.n_groups + grp] = m2;
      }
    }
  };
  using std::integral_constant;
#pragma unroll 1
  for (int t = 0; t < n_wide; t += 2) {
    if (2 * ((int64_t)blockIdx.x * n_wide + t) * kWaves >= a.T) break;
    tile(t, integral_constant<int, 0>{});
    if (t + 1 >= n_wide || 2 * ((int64_t)blockIdx.x * n_wide + t + 1) * kWaves >= a.T) break;
    tile(t + 1, integral_constant<int, 1>{});
  }
}

// band[q] = 2 delta' of query q: emit()'s level-1 bound (the pieces this query and the worst centroid actually drop, the
// fp32 accumulation of the MFMA terms, the subnormal pieces, and the exact chain's own rounding), in f' units;
// +inf when the queries or the centroids do not fit the fp16 scale (the select then evaluates the query exactly)
__device__ __forceinline__ void probe_band(const ProbeSplitOut& po, float s, float n2c, float n2r, float n2m, float& band,
                                           float& qscale) {
  const float cn = sqrtf(__uint_as_float(po.cmax2_bits[0])), cnr = sqrtf(__uint_as_float(po.cmax2_bits[1]));
  const float c2 = sqrtf(__uint_as_float(po.cmax2_bits[2]));
  const float an = sqrtf(n2c), anr = sqrtf(n2r) * s;
  const float t1 = an + cn, t2 = anr + cnr * s;
  const float a2 = sqrtf(n2m);
  const float dropped = a2 * (2.002f * cn + c2) + 1.001f * an * c2;
  float delta = 1.26f * (dropped + po.eps * t1 * t1 + po.eta * (2.f * cn + an) + po.eps_exact * t2 * t2);
  if (po.cflag[0] != 0 || !(delta < 3.0e38f)) delta = INFINITY;  // (a query beyond the scale: n2c = inf -> delta = inf)
  // the fast values are STORED as fp16 of f' x 2^-e with |f'| <= (|a'| + |c'|max)^2 = t1^2 < 2^(e + 15): half the bytes
  // of the matrix the select reads (and the sims kernel's time is its write).  The rounding of the stored values is
  // the select kernel's to add to the band: it knows how large the values near the top of the row are
  float sc = 0.f;
  const float b = t1 * t1;
  if (delta < INFINITY && b > 0.f) {
    const int e = ilogbf(b) - 14;
    if (e > -100 && e < 100) sc = ldexpf(1.f, -e);
  }
  if (!(sc > 0.f)) {  // (all-zero or astronomically scaled data: evaluated exactly)
    sc = 1.f;
    if (b > 0.f) delta = INFINITY;
  }
  qscale = sc;
  band = 2.f * delta * sc;  // in STORED units
}

// split_kernel for a search batch.  There a lane walks all of its point's dimensions, 32 at a time: four dependent
// rounds of strided loads, and 10 000 queries are 40 blocks -- 16 us of latency on a mostly idle chip.  Here a block is
// 64 queries x 4 waves and WAVE w takes k-quarter w: one round of loads per wave, all in flight together.  The raw
// values also go to LDS, from which wave 0 sums |x|^2 as the exact kernels do (one fma chain over ascending k -- the
// one quantity here whose rounding is part of the result) and derives the band; |a'|^2 and |a' - ah|^2 only enter
// bounds and are summed per quarter.  No norms are written: nothing on the probe's path reads them.
__global__ __launch_bounds__(256) void probe_split_kernel(const float* __restrict__ A, const float* __restrict__ mu,
                                                         const float* __restrict__ scale, u32x4* __restrict__ hi,
                                                         u32x4* __restrict__ mid, int d, int64_t m, int64_t T, int KS,
                                                         ProbeSplitOut po) {
  __shared__ float xs[128 * 64];      // [k][query]
  __shared__ float part[4][2][64];    // per quarter: |a'|^2, |a' - ah|^2
  __shared__ int bad_s[4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int64_t tile = i >> 5;
  const int l31 = (int)(i & 31);
  const bool iv = i < m;
  const int Q = (KS + 1) / 2;
  const float s = scale[0];
  if (w < Q) {
    const int q = w;
    const float* Ab = A + (iv ? i : 0);
    float x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int k = 32 * q + j;
      x[j] = (iv && k < d) ? Ab[(int64_t)k * m] : 0.f;
    }
    if (iv) {
      float4* xr = reinterpret_cast<float4*>(po.xt + i * po.xt_stride + 32 * q);
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (32 * q + 4 * c < po.xt_stride) xr[c] = make_float4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
    }
    float n2c = 0.f, n2m = 0.f;
    int bad = 0;
    const int64_t fo = ((tile * Q) + q) * 128 + l31 * 4;  // in 16-byte chunks
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f16x8 h, mm;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * q + 8 * c + j;
        const float xv = x[8 * c + j];
        xs[k * 64 + lane] = xv;
        const float a = (iv && k < d) ? (xv - mu[k]) * s : 0.f;
        bad |= !(fabsf(a) < 16384.f);
        const _Float16 hh = (_Float16)a;
        const float r = a - (float)hh;
        h[j] = hh;
        mm[j] = (_Float16)r;
        n2c = fmaf(a, a, n2c);
        n2m = fmaf(r, r, n2m);
      }
      if (tile < T) {
        hi[fo + c] = __builtin_bit_cast(u32x4, h);
        mid[fo + c] = __builtin_bit_cast(u32x4, mm);
      }
    }
    part[q][0][lane] = n2c;
    part[q][1][lane] = n2m;
    bad_s[q][lane] = bad;
  }
  __syncthreads();
  if (w == 0 && iv) {
    float n2r = 0.f, n2c = 0.f, n2m = 0.f;
    int bad = 0;
#pragma unroll 16
    for (int k = 0; k < 32 * Q; ++k) {
      const float xv = xs[k * 64 + lane];
      n2r = fmaf(xv, xv, n2r);
    }
    for (int q = 0; q < Q; ++q) {
      n2c += part[q][0][lane];
      n2m += part[q][1][lane];
      bad |= bad_s[q][lane];
    }
    // (the quarter sums round differently from one chain: a few ulps, under the bounds' own 1.001 factors)
    float band, qs;
    probe_band(po, s, bad ? INFINITY : n2c * 1.000001f, n2r, n2m * 1.000001f, band, qs);
    po.q2[i] = n2r;
    po.band[i] = band;
    po.qscale[i] = qs;
  }
}

// the centroids as rows, and |C|^2 as the exact kernels sum it (ascending k, fma)
__global__ __launch_bounds__(256) void probe_rows_kernel(const float* __restrict__ C, float* __restrict__ ct,
                                                        float* __restrict__ c2, int d, int n_cells) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
  float sq = 0.f;
  for (int k0 = 0; k0 < d; k0 += 32) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + ty + 8 * r, c = c0 + tx;
      tile[ty + 8 * r][tx] = (k < d && c < n_cells) ? C[(int64_t)k * n_cells + c] : 0.f;
    }
    __syncthreads();
    if (ty == 0) {  // (one thread per cell: the chain is sequential in k)
#pragma unroll
      for (int kk = 0; kk < 32; ++kk)
        if (k0 + kk < d) sq = fmaf(tile[kk][tx], tile[kk][tx], sq);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + ty + 8 * r, k = k0 + tx;
      if (c < n_cells && k < d) ct[(int64_t)c * d + k] = tile[tx][ty + 8 * r];
    }
    __syncthreads();
  }
  if (ty == 0 && c0 + tx < n_cells) c2[c0 + tx] = sq;
}

// Everything that depends on the centroids alone -- mean, scale (from the CENTROIDS' range, one bit of headroom: a
// query beyond it gets an infinite norm from split_kernel and is evaluated exactly), fp16 fragments, row copies, |C|^2
// -- is prepared once per codebook (tpq_ivfpq_coarse_probe_prepare) or, without a prepared block, per call.
struct ProbePrepared {
  int KS, chunks;
  size_t mu_off, scale_off, cflag_off, maxbits_off, cmax_off, frags_off, ct_off, c2_off, total;
};
static int probe_ks(int d) { return d <= 32 ? 2 : (d <= 64 ? 4 : 8); }
static ProbePrepared probe_prepared_layout(int d, int n_cells) {
  ProbePrepared L;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  L.KS = probe_ks(d);
  L.chunks = (n_cells + 255) / 256;
  L.mu_off = 0;
  L.scale_off = (size_t)kMu * 4;
  L.cflag_off = L.scale_off + 4;      // (also the "flag" of maxabs / scale: non-finite centroids)
  L.maxbits_off = L.cflag_off + 4;
  L.cmax_off = L.maxbits_off + 4;
  L.frags_off = up(L.cmax_off + 4 * kCm);
  L.ct_off = up(L.frags_off + (size_t)L.chunks * 8 * (2 * L.KS + 1) * 1024);
  L.c2_off = up(L.ct_off + (size_t)n_cells * d * 4);
  L.total = up(L.c2_off + (size_t)n_cells * 4);
  return L;
}
struct ProbeLayout {
  PrepLayout P;
  ProbePrepared C;
  int KS, n_groups, gshift;
  int xt_stride;
  size_t prep_off, flag_off, sims_off, gmax_off, band_off, qscale_off, q2_off, xt_off, prepared_off, total;
};
static int probe_gshift(int n_cells) { return n_cells <= 8192 ? 5 : (n_cells <= 16384 ? 6 : 7); }
static ProbeLayout probe_layout(int d, int nq, int n_cells) {
  ProbeLayout L;
  L.C = probe_prepared_layout(d, n_cells);
  L.KS = L.C.KS;
  L.P = prep_layout(1, 16 * L.KS, nq);
  // group maxima: of 32 cells up to 8 192 cells, of 64 up to 16 384 (<= 256 groups, which the select prefetches whole;
  // its direct list needs 2 n_probe <= groups), of 128 beyond
  L.gshift = probe_gshift(n_cells);
  L.n_groups = (n_cells + (1 << L.gshift) - 1) >> L.gshift;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  L.prep_off = 0;
  L.flag_off = up(L.P.total);
  L.sims_off = L.flag_off + 256;
  L.gmax_off = up(L.sims_off + (size_t)nq * n_cells * 2);
  L.band_off = up(L.gmax_off + (size_t)nq * L.n_groups * 4);
  L.qscale_off = up(L.band_off + (size_t)nq * 4);
  L.q2_off = up(L.qscale_off + (size_t)nq * 4);
  L.xt_stride = (d + 3) / 4 * 4;
  L.xt_off = up(L.q2_off + (size_t)nq * 4);
  L.prepared_off = up(L.xt_off + (size_t)nq * L.xt_stride * 4);   // (used when the caller passes no prepared block)
  L.total = L.prepared_off + L.C.total;
  return L;
}

template <int KS>
static int run_probe_prepare(const float* centroids, int d, int n_cells, char* prepared, const ProbePrepared& C,
                             hipStream_t st) {
  float* mu = reinterpret_cast<float*>(prepared + C.mu_off);
  float* scale = reinterpret_cast<float*>(prepared + C.scale_off);
  int* cflag = reinterpret_cast<int*>(prepared + C.cflag_off);
  unsigned* maxbits = reinterpret_cast<unsigned*>(prepared + C.maxbits_off);
  unsigned* cmax = reinterpret_cast<unsigned*>(prepared + C.cmax_off);
  int rc = check_hip(hipMemsetAsync(prepared, 0, C.frags_off, st), "coarse_probe_prepare memset");
  if (rc) return rc;
  hipLaunchKernelGGL(mu_kernel, dim3(d, 1), dim3(256), 0, st, centroids, mu, d, n_cells);
  TPQ_LAUNCH_CHECK("lloyd mu_kernel");
  int chunks = (int)(4096 / (int64_t)d);
  if (chunks < 1) chunks = 1;
  if ((int64_t)chunks * 4096 > n_cells) chunks = (n_cells + 4095) / 4096;
  hipLaunchKernelGGL(maxabs_kernel, dim3(chunks, d, 1), dim3(256), 0, st, centroids, mu, maxbits, cflag, d,
                     (int64_t)n_cells, 1);
  TPQ_LAUNCH_CHECK("lloyd maxabs_kernel");
  hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(64), 0, st, maxbits, cflag, scale, 1, 1);
  TPQ_LAUNCH_CHECK("lloyd scale_kernel");
  hipLaunchKernelGGL(cprep_kernel, dim3(8 * C.chunks, 1), dim3(64), 0, st, centroids, mu, scale,
                     reinterpret_cast<u32x4*>(prepared + C.frags_off), cmax, cflag, d, n_cells, KS);
  TPQ_LAUNCH_CHECK("lloyd cprep_kernel");
  hipLaunchKernelGGL(probe_rows_kernel, dim3((n_cells + 31) / 32), dim3(256), 0, st, centroids,
                     reinterpret_cast<float*>(prepared + C.ct_off), reinterpret_cast<float*>(prepared + C.c2_off), d,
                     n_cells);
  TPQ_LAUNCH_CHECK("probe_rows_kernel");
  return TPQ_OK;
}

template <int KS>
static int run_probe_sims(const float* query, const char* prepared, int d, int nq, int n_cells, char* ws,
                          const ProbeLayout& L, ProbeFastBuffers* out, hipStream_t st) {
  const PrepLayout& P = L.P;
  const ProbePrepared& C = L.C;
  char* p = ws + L.prep_off;
  int* flag = reinterpret_cast<int*>(ws + L.flag_off);   // (queries beyond the scale carry it in their norm)
  _Float16* sims = reinterpret_cast<_Float16*>(ws + L.sims_off);
  float* gmax = reinterpret_cast<float*>(ws + L.gmax_off);
  float* band = reinterpret_cast<float*>(ws + L.band_off);
  float* qscale = reinterpret_cast<float*>(ws + L.qscale_off);
  const float* mu = reinterpret_cast<const float*>(prepared + C.mu_off);
  const float* scale = reinterpret_cast<const float*>(prepared + C.scale_off);
  const int* cflag = reinterpret_cast<const int*>(prepared + C.cflag_off);
  const unsigned* cmax = reinterpret_cast<const unsigned*>(prepared + C.cmax_off);
  const u32x4* frags = reinterpret_cast<const u32x4*>(prepared + C.frags_off);
  float* q2 = reinterpret_cast<float*>(ws + L.q2_off);
  float* xt = reinterpret_cast<float*>(ws + L.xt_off);
  const ProbeSplitOut po{xt, q2, band, qscale, cmax, cflag, level_eps(KS, 16 * KS, 1), (float)(d + 4) / 16777216.0f,
                         sqrtf((float)(16 * KS)) / 8192.0f, L.xt_stride};
  hipLaunchKernelGGL(probe_split_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(256), 0, st, query, mu, scale,
                     reinterpret_cast<u32x4*>(p + P.hi_off), reinterpret_cast<u32x4*>(p + P.mid_off), d, (int64_t)nq, P.T, KS,
                     po);
  TPQ_LAUNCH_CHECK("probe_split_kernel");
  const size_t lds = (size_t)8 * (KS + 1) * 1024;
  auto kernel = L.gshift == 5 ? probe_sims_kernel<KS, 5> : (L.gshift == 6 ? probe_sims_kernel<KS, 6> : probe_sims_kernel<KS, 7>);
  int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds), "probe_sims_kernel attr");
  if (rc) return rc;
  // wide tiles (64 queries) per wave: as few as it takes to put >= ~1 000 blocks on the chip
  const int64_t wide = (P.T + 1) / 2;
  int n_wide = (int)((wide * C.chunks) / ((int64_t)kWaves * 1024));
  n_wide = n_wide < 1 ? 1 : (n_wide > kWide ? kWide : n_wide);
  // the block's sims rows are ONE buffer resource addressed with 32-bit offsets (probe_sims_kernel): its
  // rows x n_cells x 2 bytes must stay below the out-of-range sentinel 0x7ffffff0 (at 262 144 cells a block of
  // 8 192 rows was 4 GiB: num_records truncated to 0, row offsets wrapped).  lloyd_probe_supported() keeps one
  // wide tile per wave inside the range; here the tiles per wave are cut to what fits.
  const int64_t row_bytes = (int64_t)n_cells * 2, rows_per_wide = (int64_t)kWaves * 64;
  const int64_t fit = (int64_t)0x7ffffff0 / (row_bytes * rows_per_wide);
  if (fit < 1) {
    set_error("probe_sims: %d cells: one block's rows exceed the 2 GiB buffer resource", n_cells);
    return TPQ_ERR_UNSUPPORTED;
  }
  n_wide = n_wide > fit ? (int)fit : n_wide;
  const int64_t per_block = (int64_t)kWaves * n_wide;
  ProbeSimsArgs pa{reinterpret_cast<const u32x4*>(p + P.hi_off), frags, sims, qscale, gmax, nq, n_cells, L.n_groups, n_wide,
                   P.T, 8 * (2 * KS + 1) * 64};
  hipLaunchKernelGGL(kernel, dim3((unsigned)((wide + per_block - 1) / per_block), C.chunks), dim3(kWaves * 64), lds, st,
                     pa);
  TPQ_LAUNCH_CHECK("probe_sims_kernel");
  *out = ProbeFastBuffers{sims, gmax, band, qscale, xt, q2, L.xt_stride, reinterpret_cast<const float*>(prepared + C.ct_off),
                          reinterpret_cast<const float*>(prepared + C.c2_off), L.n_groups, L.gshift};
  return TPQ_OK;
}

}  // namespace lloyd

// hooks for tpq_ivfpq_coarse_probe (select.hip, probe_fast.h): euclidean, d <= 128, whole 16-byte pieces per row
int lloyd_probe_supported(int d, int nq, int n_cells) {
  // d % 4: probe_select_fast_kernel reads the centroid rows (stride d floats) as float4.
  // n_cells <= 2^20: the 512 rows of one wide tile per wave (kWaves x 64) x n_cells x 2 bytes must fit the 32-bit
  // buffer resource of probe_sims_kernel (run_probe_sims cuts the tiles per wave to what fits).
  if (!(d >= 4 && d <= 128 && (d & 3) == 0 && nq >= 1 && n_cells >= 256 && (n_cells & 31) == 0 && n_cells <= (1 << 20)))
    return 0;
  if ((int64_t)n_cells * 2 * lloyd::kWaves * 64 > (int64_t)0x7ffffff0) return 0;
  return (int64_t)nq * n_cells < (1LL << 36) ? 1 : 0;
}
int lloyd_probe_groups(int n_cells) {
  const int gs = lloyd::probe_gshift(n_cells);
  return (n_cells + (1 << gs) - 1) >> gs;
}
size_t lloyd_probe_workspace_bytes(int d, int nq, int n_cells) {
  return lloyd_probe_supported(d, nq, n_cells) ? lloyd::probe_layout(d, nq, n_cells).total : 0;
}
size_t lloyd_probe_prepared_bytes(int d, int n_cells) {
  return lloyd_probe_supported(d, 1, n_cells) ? lloyd::probe_prepared_layout(d, n_cells).total : 0;
}
int lloyd_probe_prepare(const float* centroids, int d, int n_cells, char* prepared, hipStream_t st) {
  const lloyd::ProbePrepared C = lloyd::probe_prepared_layout(d, n_cells);
  switch (C.KS) {
    case 2: return lloyd::run_probe_prepare<2>(centroids, d, n_cells, prepared, C, st);
    case 4: return lloyd::run_probe_prepare<4>(centroids, d, n_cells, prepared, C, st);
    default: return lloyd::run_probe_prepare<8>(centroids, d, n_cells, prepared, C, st);
  }
}
int lloyd_probe_sims(const float* query, const float* centroids, const void* prepared, int d, int nq, int n_cells,
                     char* ws, ProbeFastBuffers* out, hipStream_t st) {
  const lloyd::ProbeLayout L = lloyd::probe_layout(d, nq, n_cells);
  const char* prep = reinterpret_cast<const char*>(prepared);
  if (!prep) {  // no prepared block: prepare into the workspace, for this call
    int rc = lloyd_probe_prepare(centroids, d, n_cells, ws + L.prepared_off, st);
    if (rc) return rc;
    prep = ws + L.prepared_off;
  }
  switch (L.KS) {
    case 2: return lloyd::run_probe_sims<2>(query, prep, d, nq, n_cells, ws, L, out, st);
    case 4: return lloyd::run_probe_sims<4>(query, prep, d, nq, n_cells, ws, L, out, st);
    default: return lloyd::run_probe_sims<8>(query, prep, d, nq, n_cells, ws, L, out, st);
  }
}
int lloyd_assign_supported(int d, int64_t m, int n, int route) {
  if (!(d >= 1 && d <= 128 && n >= 1 && n <= (1 << 24) && m >= 1 && m < (1LL << 31))) return 0;
  if (TPQ_AB_ENV("TPQ_COARSE_ASSIGN_OLD")) return 0;  // (A/B: the two-piece bf16 selection of assign_fast.hip)
  // below ~4 096 centroids the per-call preparation (max-abs + split of the points, the fold of the
  // chunks) costs more than the lighter sweep saves: 128 x 2 048: 1.66 vs 1.48 ms, 128 x 4 096: 2.37 vs 2.76,
  // 128 x 16 384: 8.5 vs 11.0, 64 x 16 384: 4.7 vs 6.0, 128 x 65 536: 39.9 vs 47.4 (1 M points)
  // (route == TPQ_ASSIGN_ROUTE_CASCADE: the caller asks for the cascade whatever the size -- tests, tuning)
  if (route != TPQ_ASSIGN_ROUTE_CASCADE && n < 4096) return 0;
  const lloyd::PrepLayout P = lloyd::prep_layout(1, d, m);
  return (P.T * ((P.KS + 1) / 2) * 2048 <= 0x7fffffffLL && (int64_t)d * m * 4 <= 0x7fffffffLL) ? 1 : 0;
}
// 128 < d <= 1024: the GEMM-shaped cascade (euclidean)
int lloyd_wide_supported(int d, int64_t m, int n) {
  return (d > 128 && d <= 1024 && n >= 1 && n <= (1 << 22) && m >= 1 && m < (1LL << 28)) ? 1 : 0;
}
size_t lloyd_assign_workspace_bytes(int d, int64_t m, int n) {
  return d > 128 ? lloyd::wide_layout(d, m, n).total : lloyd::assign_layout(d, m, n).total;
}
size_t lloyd_assign_count_offset(int d, int64_t m, int n) {  // wide: the points with an exact step (candidates)
  if (d > 128) return lloyd::wide_layout(d, m, n).count1_off;
  const lloyd::AssignLayout L = lloyd::assign_layout(d, m, n);
  // (chunked problems on the candidate route: the points that got an exact step on their candidates)
  const bool cand = L.chunks > 1 && n <= (1 << 22) &&
                    !(TPQ_AB_ENV("TPQ_COARSE_ASSIGN_CAND") && atoi(TPQ_AB_ENV("TPQ_COARSE_ASSIGN_CAND")) == 0);
  return cand ? L.count1_off : L.count2_off;
}
int lloyd_assign(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n, int euclid,
                 char* ws, hipStream_t st) {
  if (d > 128) return lloyd::run_wide(A, B, vals, inds, d, m, n, euclid, ws, lloyd::wide_layout(d, m, n), st);
  const lloyd::AssignLayout L = lloyd::assign_layout(d, m, n);
  switch (L.KS) {
    case 1: return lloyd::run_assign<1>(A, B, vals, inds, d, m, n, ws, L, st);
    case 2: return lloyd::run_assign<2>(A, B, vals, inds, d, m, n, ws, L, st);
    case 3: return lloyd::run_assign<3>(A, B, vals, inds, d, m, n, ws, L, st);
    case 4: return lloyd::run_assign<4>(A, B, vals, inds, d, m, n, ws, L, st);
    case 5: return lloyd::run_assign<5>(A, B, vals, inds, d, m, n, ws, L, st);
    case 6: return lloyd::run_assign<6>(A, B, vals, inds, d, m, n, ws, L, st);
    case 7: return lloyd::run_assign<7>(A, B, vals, inds, d, m, n, ws, L, st);
    default: return lloyd::run_assign<8>(A, B, vals, inds, d, m, n, ws, L, st);
  }
}
}  // namespace tpq

using namespace tpq;

extern "C" int tpq_lloyd_supported(int l, int d, int64_t m, int n) {
  if (!(l >= 1 && l <= 65535 && d >= 1 && d <= 64 && n >= 1 && n <= 256 && m >= 1 && m < (1LL << 31))) return 0;
  const lloyd::PrepLayout L = lloyd::prep_layout(l, d, m);
  return (L.T * ((L.KS + 1) / 2) * 2048 <= 0x7fffffffLL && (int64_t)d * m * 4 <= 0x7fffffffLL) ? 1 : 0;
}

extern "C" size_t tpq_lloyd_prepared_bytes(int l, int d, int64_t m) {
  if (l < 1 || d < 1 || d > 64 || m < 1) return 0;
  return lloyd::prep_layout(l, d, m).total;
}

extern "C" int tpq_lloyd_prepare(const float* data, const float* centroids0, void* prepared, size_t prepared_bytes,
                                 int l, int d, int64_t m, int n, tpq_stream_t stream) {
  TPQ_REQUIRE(data && centroids0 && prepared, "lloyd_prepare: null pointer");
  if (!tpq_lloyd_supported(l, d, m, n)) {
    set_error("lloyd_prepare: shape l=%d d=%d m=%lld n=%d not supported (d <= 64, n <= 256, slices < 2 GiB)", l, d,
              (long long)m, n);
    return TPQ_ERR_UNSUPPORTED;
  }
  const lloyd::PrepLayout L = lloyd::prep_layout(l, d, m);
  TPQ_REQUIRE(prepared_bytes >= L.total, "lloyd_prepare: prepared block of %zu bytes needed", L.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* p = reinterpret_cast<char*>(prepared);
  float* mu = reinterpret_cast<float*>(p + L.mu_off);
  float* scale = reinterpret_cast<float*>(p + L.scale_off);
  int* flag = reinterpret_cast<int*>(p + L.flag_off);
  unsigned* maxbits = reinterpret_cast<unsigned*>(p + L.maxbits_off);
  int rc = check_hip(hipMemsetAsync(p + L.mu_off, 0, L.total - L.mu_off, st), "lloyd_prepare memset");
  if (rc) return rc;
  hipLaunchKernelGGL(lloyd::mu_kernel, dim3(d, l), dim3(256), 0, st, centroids0, mu, d, n);
  TPQ_LAUNCH_CHECK("lloyd mu_kernel");
  int chunks = (int)(4096 / ((int64_t)l * d));
  if (chunks < 1) chunks = 1;
  if ((int64_t)chunks * 4096 > m) chunks = (int)((m + 4095) / 4096);
  // the scale is a power of two: read a sixteenth of a large problem for it (every sixteenth 4-KiB run of each
  // row) and leave one bit of headroom; split_kernel flags the sub-problem whose data exceed it after all
  const int sample = m >= (1 << 18) ? 16 : 1;
  hipLaunchKernelGGL(lloyd::maxabs_kernel, dim3(chunks, d, l), dim3(256), 0, st, data, mu, maxbits, flag, d, m,
                     sample);
  TPQ_LAUNCH_CHECK("lloyd maxabs_kernel");
  hipLaunchKernelGGL(lloyd::scale_kernel, dim3((l + 63) / 64), dim3(64), 0, st, maxbits, flag, scale, l,
                     sample > 1 ? 1 : 0);
  TPQ_LAUNCH_CHECK("lloyd scale_kernel");
  hipLaunchKernelGGL(lloyd::split_kernel, dim3((unsigned)((L.T + 7) / 8), l), dim3(256), 0, st, data, mu, scale,
                     reinterpret_cast<lloyd::u32x4*>(p + L.hi_off), reinterpret_cast<lloyd::u32x4*>(p + L.mid_off),
                     reinterpret_cast<float2*>(p + L.norms_off), flag, d, m, L.T, L.KS);
  TPQ_LAUNCH_CHECK("lloyd split_kernel");
  return TPQ_OK;
}

extern "C" size_t tpq_lloyd_step_workspace_bytes(int l, int d, int64_t m, int n) {
  if (!tpq_lloyd_supported(l, d, m, n)) return 0;
  return lloyd::step_layout(l, d, m, n).total;
}

// diagnostics: byte offsets of the int32 [l] counts of points left undecided by level 1 / level 2
extern "C" size_t tpq_lloyd_step_count_offset(int l, int d, int64_t m, int n, int level) {
  if (!tpq_lloyd_supported(l, d, m, n)) return 0;
  const lloyd::StepLayout L = lloyd::step_layout(l, d, m, n);
  return level == 1 ? L.count_off : L.count2_off;
}

extern "C" int tpq_lloyd_step(const float* data, const void* prepared, const float* centroids, float* new_centroids,
                              float* vals, int64_t* inds, int l, int d, int64_t m, int n, void* workspace,
                              size_t workspace_bytes, tpq_stream_t stream) {
  TPQ_REQUIRE(data && prepared && centroids && inds, "lloyd_step: null pointer");
  if (!tpq_lloyd_supported(l, d, m, n)) {
    set_error("lloyd_step: shape l=%d d=%d m=%lld n=%d not supported", l, d, (long long)m, n);
    return TPQ_ERR_UNSUPPORTED;
  }
  const lloyd::PrepLayout P = lloyd::prep_layout(l, d, m);
  const lloyd::StepLayout L = lloyd::step_layout(l, d, m, n);
  TPQ_REQUIRE(workspace && workspace_bytes >= L.total, "lloyd_step: workspace of %zu bytes needed", L.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const char* p = reinterpret_cast<const char*>(prepared);
  char* ws = reinterpret_cast<char*>(workspace);
  const float* mu = reinterpret_cast<const float*>(p + P.mu_off);
  const float* scale = reinterpret_cast<const float*>(p + P.scale_off);
  lloyd::u32x4* frags = reinterpret_cast<lloyd::u32x4*>(ws + L.frags_off);
  unsigned* cmax = reinterpret_cast<unsigned*>(ws + L.cmax_off);
  int* count = reinterpret_cast<int*>(ws + L.count_off);
  int* cflag = reinterpret_cast<int*>(ws + L.cflag_off);
  int* count2 = reinterpret_cast<int*>(ws + L.count2_off);
  int* list = reinterpret_cast<int*>(ws + L.list_off);
  int* list2 = reinterpret_cast<int*>(ws + L.list2_off);
  int rc = check_hip(hipMemsetAsync(ws + L.cmax_off, 0, L.list_off - L.cmax_off, st), "lloyd_step memset");
  if (rc) return rc;
  const int KS = P.KS;
  hipLaunchKernelGGL(lloyd::cprep_kernel, dim3(8, l), dim3(64), 0, st, centroids, mu, scale, frags, cmax, cflag, d, n,
                     KS);
  TPQ_LAUNCH_CHECK("lloyd cprep_kernel");
  lloyd::StepArgs sa{reinterpret_cast<const lloyd::u32x4*>(p + P.hi_off),
                     reinterpret_cast<const lloyd::u32x4*>(p + P.mid_off),
                     reinterpret_cast<const float2*>(p + P.norms_off),
                     frags, cmax, scale, reinterpret_cast<const int*>(p + P.flag_off), cflag, inds, vals,
                     nullptr, nullptr, list, count, (int)m, P.T,
                     0.f, (float)(d + 4) / 16777216.0f, sqrtf((float)d) / 8192.0f, 1, nullptr, 0,
                     nullptr, nullptr, 0};
  switch (KS) {
    case 1: rc = lloyd::run_levels<1>(sa, l, d, list2, count2, st); break;
    case 2: rc = lloyd::run_levels<2>(sa, l, d, list2, count2, st); break;
    case 3: rc = lloyd::run_levels<3>(sa, l, d, list2, count2, st); break;
    default: rc = lloyd::run_levels<4>(sa, l, d, list2, count2, st); break;
  }
  if (rc) return rc;
  rc = launch_max_sim_list(data, centroids, vals, inds, l, d, (int)m, n, 1, list2, count2, nullptr, nullptr, 0, st);
  if (rc) return rc;
  if (new_centroids) {
    if (TPQ_AB_ENV("TPQ_LL_OLD_UPDATE"))  // (A/B: the fp32-data update of kmeans.hip)
      return tpq_compute_centroids(data, inds, new_centroids, l, d, m, n, ws + L.upd_off,
                                   tpq_compute_centroids_workspace_bytes(l, d, n), stream);
    float* sums = reinterpret_cast<float*>(ws + L.upd_off);
    float* counts = sums + (size_t)l * d * n;
    rc = check_hip(hipMemsetAsync(sums, 0, tpq_compute_centroids_workspace_bytes(l, d, n), st), "lloyd_step memset");
    if (rc) return rc;
    lloyd::UpdArgs ua{reinterpret_cast<const lloyd::u32x4*>(p + P.hi_off),
                      reinterpret_cast<const lloyd::u32x4*>(p + P.mid_off), inds,
                      reinterpret_cast<const int*>(p + P.flag_off), sums, counts, d, n, (int)m, P.T};
    switch (KS) {
      case 1: return lloyd::run_update<1>(ua, data, mu, scale, new_centroids, l, st);
      case 2: return lloyd::run_update<2>(ua, data, mu, scale, new_centroids, l, st);
      case 3: return lloyd::run_update<3>(ua, data, mu, scale, new_centroids, l, st);
      default: return lloyd::run_update<4>(ua, data, mu, scale, new_centroids, l, st);
    }
  }
  return TPQ_OK;
}
