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

// A candidate is ONE 64-bit key: high word = order-preserving image of the fp32 value, low word
// = ~index.  "better" (larger value, ties -> smaller index) is a single unsigned 64-bit compare,
// which is what keeps a bitonic compare-exchange at ~6 VALU instead of ~14.
// (-0.0f and +0.0f would order differently from the float compare; callers canonicalise with
// v + 0.0f -- a sum started at +0.f is never -0.)
__device__ __forceinline__ unsigned f2key(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
struct Key {
  unsigned hi, lo;
};
__device__ __forceinline__ Key make_key(float v, int idx) { return Key{f2key(v), ~(unsigned)idx}; }
__device__ __forceinline__ Key pad_key() { return make_key(-INFINITY, kPadIdx); }
__device__ __forceinline__ float key_value(Key k) { return key2f(k.hi); }
__device__ __forceinline__ int key_index(Key k) { return (int)~k.lo; }
__device__ __forceinline__ unsigned long long key_u64(Key k) {
  return ((unsigned long long)k.hi << 32) | k.lo;
}
__device__ __forceinline__ bool key_better(Key a, Key b) { return key_u64(a) > key_u64(b); }

__device__ __forceinline__ bool kv_better(float av, int ai, float bv, int bi) {
  return (av > bv) || (av == bv && ai < bi);
}

__device__ __forceinline__ float readlane_f(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
__device__ __forceinline__ int readlane_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ Key readlane_key(Key k, int l) {
  return Key{(unsigned)__builtin_amdgcn_readlane((int)k.hi, l),
             (unsigned)__builtin_amdgcn_readlane((int)k.lo, l)};
}

// value of lane (lane ^ J) without an LDS round trip where the ISA allows it:
//   J = 1, 2: DPP quad_perm; J = 4: row_half_mirror o quad_perm(3,2,1,0) (i^7 then ^3);
//   J = 8: DPP row_ror:8; J = 16: ds_swizzle bit-mask mode (xor 0x10, no address VGPR);
//   J = 32: ds_bpermute.  (hipcc lowers __shfl_xor to ds_bpermute_b32 for every J.)
// (The DPP moves carry NO old value -- mov_dpp, bound_ctrl set: every lane of these permutations has a source lane inside
// its row, and the networks run with all 64 lanes active.  Written as update_dpp(x, x, ...) the destination was tied to x
// and hipcc copied both words of the key before every exchange: 7 VALU per compare-exchange, 5 now -- a flush is 45 of
// them, and a short list spends as much on its flushes as on its look-ups)
template <int J>
__device__ __forceinline__ int xor_lane_i(int x) {
  if constexpr (J == 1) {
    return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  } else if constexpr (J == 2) {
    return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  } else if constexpr (J == 4) {
    const int t = __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true);  // row_half_mirror
    return __builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, true);          // quad_perm [3,2,1,0]
  } else if constexpr (J == 8) {
    return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);  // row_ror:8
  } else if constexpr (J == 16) {
    return __builtin_amdgcn_ds_swizzle(x, 0x401F);  // and 0x1f, or 0, xor 0x10
  } else {
    return __shfl_xor(x, J, 64);
  }
}

// one compare-exchange step with the lane at distance J (xor)
template <int J>
__device__ __forceinline__ void cmpx(Key& k, bool take_better) {
  const Key o{(unsigned)xor_lane_i<J>((int)k.hi), (unsigned)xor_lane_i<J>((int)k.lo)};
  const bool keep = (key_better(k, o) == take_better);
  k.hi = keep ? k.hi : o.hi;
  k.lo = keep ? k.lo : o.lo;
}

// bitonic (any rotation of up-then-down) 64-sequence -> sorted, lane 0 = best
__device__ __forceinline__ void merge64_desc(Key& k) {
  const int lane = lane_id();
  cmpx<32>(k, (lane & 32) == 0);
  cmpx<16>(k, (lane & 16) == 0);
  cmpx<8>(k, (lane & 8) == 0);
  cmpx<4>(k, (lane & 4) == 0);
  cmpx<2>(k, (lane & 2) == 0);
  cmpx<1>(k, (lane & 1) == 0);
}

template <int K, int J>
__device__ __forceinline__ void sort_step(Key& k, int lane) {
  const bool desc = (K == 64) ? true : ((lane & K) == 0);
  cmpx<J>(k, ((lane & J) == 0) == desc);
}

// arbitrary 64 keys (one per lane) -> sorted, lane 0 = best
__device__ __forceinline__ void sort64_desc(Key& k) {
  const int lane = lane_id();
  sort_step<2, 1>(k, lane);
  sort_step<4, 2>(k, lane);
  sort_step<4, 1>(k, lane);
  sort_step<8, 4>(k, lane);
  sort_step<8, 2>(k, lane);
  sort_step<8, 1>(k, lane);
  sort_step<16, 8>(k, lane);
  sort_step<16, 4>(k, lane);
  sort_step<16, 2>(k, lane);
  sort_step<16, 1>(k, lane);
  sort_step<32, 16>(k, lane);
  sort_step<32, 8>(k, lane);
  sort_step<32, 4>(k, lane);
  sort_step<32, 2>(k, lane);
  sort_step<32, 1>(k, lane);
  sort_step<64, 32>(k, lane);
  sort_step<64, 16>(k, lane);
  sort_step<64, 8>(k, lane);
  sort_step<64, 4>(k, lane);
  sort_step<64, 2>(k, lane);
  sort_step<64, 1>(k, lane);
}

// The wave's sorted best 64*R candidates.
template <int R>
struct WaveTopK {
  Key k[R];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = pad_key();
  }

  // Fold a batch that is already sorted (lane 0 = best) into the list; losers fall off the end.
  __device__ __forceinline__ void insert_sorted(Key x) {
    const int lane = lane_id();
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
      // wave-uniform early exit: batch's best does not beat this register's worst
      if (!key_better(readlane_key(x, 0), readlane_key(k[r], 63))) {
        if (r < R - 1) k[r + 1] = x;
        return;
      }
      const Key rv{(unsigned)__shfl((int)x.hi, 63 - lane, 64),  // batch reversed: lane 0 = worst
                   (unsigned)__shfl((int)x.lo, 63 - lane, 64)};
      const bool ab = key_better(k[r], rv);
      Key h{ab ? k[r].hi : rv.hi, ab ? k[r].lo : rv.lo};  // best 64 of the 128 (bitonic)
      if (r < R - 1) {
        Key l{ab ? rv.hi : k[r].hi, ab ? rv.lo : k[r].lo};  // other 64 -> one register lower
        merge64_desc(l);
        k[r + 1] = l;
      }
      merge64_desc(h);
      x = h;
    }
    k[0] = x;
  }

  __device__ __forceinline__ void insert_unsorted(Key x) {
    sort64_desc(x);
    insert_sorted(x);
  }

  // value of the kth best (1-based kth <= 64*R); -inf while fewer than kth candidates are held
  __device__ __forceinline__ float kth_value(int kth) const {
    const int kr = (kth - 1) >> 6, kl = (kth - 1) & 63;
    float t = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r == kr) t = key2f((unsigned)__builtin_amdgcn_readlane((int)k[r].hi, kl));
    return t;
  }
};

// Threshold filter + per-wave LDS queue in front of a WaveTopK.
template <int R>
struct WaveSelector {
  WaveTopK<R> top;
  float* qv;  // [64] LDS, owned by this wave
  int* qi;    // [64]
  int qn;     // wave-uniform
  float tau;  // admission threshold: candidates with v < tau cannot reach the final top-k
  int k;
  int n_flush;  // flushes so far (wave-uniform)
  float margin; // admission slack: candidates down to tau - margin are kept (packed scan: 2*delta)

  __device__ __forceinline__ void init(float* qv_, int* qi_, int k_) {
    top.init();
    qv = qv_;
    qi = qi_;
    qn = 0;
    tau = -INFINITY;
    k = k_;
    n_flush = 0;
    margin = 0.f;
  }

  __device__ __forceinline__ void flush() {
    if (qn == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const int lane = lane_id();
    const bool act = lane < qn;
    float bv = act ? qv[lane] : -INFINITY;
    int bi = act ? qi[lane] : kPadIdx;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    qn = 0;
    ++n_flush;
    if (!(bv >= tau - margin)) {  // fell under the (possibly raised) threshold, or NaN: drop
      bv = -INFINITY;
      bi = kPadIdx;
    }
    top.insert_unsorted(make_key(bv, bi));
    tau = fmaxf(tau, top.kth_value(k));
  }

  // Large k (the packed scan's pool mode): every admitted candidate is ALSO appended to an append-only pool in
  // the caller's workspace (keys: value image, ~index); the sorted list then only has to be long enough for the
  // admission threshold, not for k.  A full pool drops the candidate and latches `n` beyond `cap` (the query is
  // then redone exactly).
  struct Pool {
    unsigned* hi;
    unsigned* lo;
    int n, cap;  // wave-uniform
  };
  __device__ __forceinline__ void push_pool(Pool& pool, bool pass, float v, int idx) {
    const unsigned long long mask = __ballot(pass);
    if (mask == 0ull) return;
    const int n = __popcll(mask);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
    if (pool.n + n <= pool.cap) {
      if (pass) {
        pool.hi[pool.n + rank] = f2key(v);
        pool.lo[pool.n + rank] = ~(unsigned)idx;
      }
      pool.n += n;
    } else {
      pool.n = pool.cap + 1;  // overflow, latched
    }
    push(pass, v, idx);
  }

  // every lane calls; lanes with pass==true enqueue (v, idx)
  __device__ __forceinline__ void push(bool pass, float v, int idx) {
    unsigned long long mask = __ballot(pass);
    if (mask == 0ull) return;
    int n = __popcll(mask);
    if (qn + n > 64) {
      flush();
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
