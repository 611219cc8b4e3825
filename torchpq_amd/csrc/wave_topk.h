// Wave-level (64-lane) running top-k for gfx950.
//
// A wave keeps its current best 64*R candidates SORTED in registers (rank e lives in
// register e/64, lane e%64), filters new candidates against a threshold, parks the
// survivors in a 64-entry LDS queue it owns, and folds a full queue into the sorted list
// with in-register bitonic networks (cross-lane moves only -- no workgroup barrier in
// steady state).  This replaces the reference's block-wide scheme of >= 9 __syncthreads
// per tile (torchpq/kernels/cuda/ivfpq_topk.cu:886-929) and its warp=32 queues in
// top32_select.cu / topk_select.cu.
//
// Order: value descending, exact ties by ascending index => a strict total order, so the
// result is unique and deterministic (the reference's network duplicates/loses ids on
// ties, ivfpq_topk.cu:50-61).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace tpq {

constexpr int kPadIdx = 0x7fffffff;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ bool kv_better(float av, int ai, float bv, int bi) {
  return (av > bv) || (av == bv && ai < bi);
}

__device__ __forceinline__ float readlane_f(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
__device__ __forceinline__ int readlane_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

// order-preserving float -> uint key (for LDS atomicMax on a shared threshold)
__device__ __forceinline__ unsigned f2key(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// one compare-exchange step with the lane at distance J (xor)
template <int J>
__device__ __forceinline__ void cmpx(float& v, int& i, bool take_better) {
  const float ov = __shfl_xor(v, J, 64);
  const int oi = __shfl_xor(i, J, 64);
  const bool keep = (kv_better(v, i, ov, oi) == take_better);
  v = keep ? v : ov;
  i = keep ? i : oi;
}

// bitonic (any rotation of up-then-down) 64-sequence -> sorted, lane 0 = best
__device__ __forceinline__ void merge64_desc(float& v, int& i) {
  const int lane = lane_id();
  cmpx<32>(v, i, (lane & 32) == 0);
  cmpx<16>(v, i, (lane & 16) == 0);
  cmpx<8>(v, i, (lane & 8) == 0);
  cmpx<4>(v, i, (lane & 4) == 0);
  cmpx<2>(v, i, (lane & 2) == 0);
  cmpx<1>(v, i, (lane & 1) == 0);
}

template <int K, int J>
__device__ __forceinline__ void sort_step(float& v, int& i, int lane) {
  const bool desc = (K == 64) ? true : ((lane & K) == 0);
  cmpx<J>(v, i, ((lane & J) == 0) == desc);
}

// arbitrary 64 values (one per lane) -> sorted, lane 0 = best
__device__ __forceinline__ void sort64_desc(float& v, int& i) {
  const int lane = lane_id();
  sort_step<2, 1>(v, i, lane);
  sort_step<4, 2>(v, i, lane);
  sort_step<4, 1>(v, i, lane);
  sort_step<8, 4>(v, i, lane);
  sort_step<8, 2>(v, i, lane);
  sort_step<8, 1>(v, i, lane);
  sort_step<16, 8>(v, i, lane);
  sort_step<16, 4>(v, i, lane);
  sort_step<16, 2>(v, i, lane);
  sort_step<16, 1>(v, i, lane);
  sort_step<32, 16>(v, i, lane);
  sort_step<32, 8>(v, i, lane);
  sort_step<32, 4>(v, i, lane);
  sort_step<32, 2>(v, i, lane);
  sort_step<32, 1>(v, i, lane);
  sort_step<64, 32>(v, i, lane);
  sort_step<64, 16>(v, i, lane);
  sort_step<64, 8>(v, i, lane);
  sort_step<64, 4>(v, i, lane);
  sort_step<64, 2>(v, i, lane);
  sort_step<64, 1>(v, i, lane);
}

// The wave's sorted best 64*R candidates.
template <int R>
struct WaveTopK {
  float v[R];
  int i[R];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = -INFINITY;
      i[r] = kPadIdx;
    }
  }

  // Fold a batch that is already sorted (lane 0 = best) into the list; losers fall off the end.
  __device__ __forceinline__ void insert_sorted(float xv, int xi) {
    const int lane = lane_id();
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
      // wave-uniform early exit: batch's best does not beat this register's worst
      const float x0v = readlane_f(xv, 0);
      const int x0i = readlane_i(xi, 0);
      const float awv = readlane_f(v[r], 63);
      const int awi = readlane_i(i[r], 63);
      if (!kv_better(x0v, x0i, awv, awi)) {
        if (r < R - 1) {
          v[r + 1] = xv;
          i[r + 1] = xi;
        }
        return;
      }
      const float rv = __shfl(xv, 63 - lane, 64);  // batch reversed: lane 0 = its worst
      const int ri = __shfl(xi, 63 - lane, 64);
      const bool ab = kv_better(v[r], i[r], rv, ri);
      float hv = ab ? v[r] : rv;  // best 64 of the 128 (bitonic)
      int hi = ab ? i[r] : ri;
      if (r < R - 1) {
        float lv = ab ? rv : v[r];  // other 64 (bitonic) -> settle one register lower
        int li = ab ? ri : i[r];
        merge64_desc(lv, li);
        v[r + 1] = lv;
        i[r + 1] = li;
      }
      merge64_desc(hv, hi);
      xv = hv;
      xi = hi;
    }
    v[0] = xv;
    i[0] = xi;
  }

  __device__ __forceinline__ void insert_unsorted(float xv, int xi) {
    sort64_desc(xv, xi);
    insert_sorted(xv, xi);
  }

  // value of the k-th best (1-based k <= 64*R); -inf while fewer than k candidates are held
  __device__ __forceinline__ float kth_value(int k) const {
    const int kr = (k - 1) >> 6, kl = (k - 1) & 63;
    float t = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r == kr) t = readlane_f(v[r], kl);
    return t;
  }
};

// Threshold filter + per-wave LDS queue in front of a WaveTopK.
// `Refine` lets the caller replace a queued candidate's value before it is ranked
// (used by the packed scan to re-evaluate survivors in the reference's summation order).
struct NoRefine {
  __device__ __forceinline__ float operator()(float v, int /*idx*/, bool /*active*/) const { return v; }
};

template <int R>
struct WaveSelector {
  WaveTopK<R> top;
  float* qv;  // [64] LDS, owned by this wave
  int* qi;    // [64]
  int qn;     // wave-uniform
  float tau;  // admission threshold: candidates with v < tau cannot reach the final top-k
  int k;
  int n_flush;  // flushes so far (wave-uniform)

  __device__ __forceinline__ void init(float* qv_, int* qi_, int k_) {
    top.init();
    qv = qv_;
    qi = qi_;
    qn = 0;
    tau = -INFINITY;
    k = k_;
    n_flush = 0;
  }

  template <class Refine>
  __device__ __forceinline__ void flush(const Refine& refine) {
    if (qn == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const int lane = lane_id();
    const bool act = lane < qn;
    float bv = act ? qv[lane] : -INFINITY;
    int bi = act ? qi[lane] : kPadIdx;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    qn = 0;
    ++n_flush;
    bv = refine(bv, bi, act);
    if (!(bv >= tau)) {  // refined value fell under the threshold (or NaN): drop
      bv = -INFINITY;
      bi = kPadIdx;
    }
    top.insert_unsorted(bv, bi);
    tau = fmaxf(tau, top.kth_value(k));
  }

  // every lane calls; lanes with pass==true enqueue (v, idx)
  template <class Refine>
  __device__ __forceinline__ void push(bool pass, float v, int idx, const Refine& refine) {
    unsigned long long mask = __ballot(pass);
    if (mask == 0ull) return;
    int n = __popcll(mask);
    if (qn + n > 64) {
      flush(refine);
    }
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                               __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
    if (pass) {
      qv[qn + rank] = v;
      qi[qn + rank] = idx;
    }
    qn += n;
  }
};

}  // namespace tpq
