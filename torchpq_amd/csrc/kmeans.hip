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
#include "common.h"

namespace tpq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMsCent = 256;  // centroids per pass (8 MFMA row tiles)
constexpr int kMsKC = 32;     // k rows staged in LDS per step

// grid (ceil(m/128), l), block 256 = 4 waves x 32 points
__global__ __launch_bounds__(256) void max_sim_kernel(const float* __restrict__ A,
                                                      const float* __restrict__ B,
                                                      float* __restrict__ vals,
                                                      int64_t* __restrict__ inds, int d, int m,
                                                      int n, int euclidean) {
  __shared__ float cs[kMsKC * kMsCent];
  __shared__ float b2s[kMsCent];
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int i = blockIdx.x * 128 + wave * 32 + l31;  // this lane's point
  const bool iv = i < m;
  const float* __restrict__ Ab = A + (int64_t)b * d * m;
  const float* __restrict__ Bb = B + (int64_t)b * d * n;

  float a2 = 0.f;
  if (euclidean && iv)
    for (int k = 0; k < d; ++k) {
      const float x = Ab[(int64_t)k * m + i];
      a2 = fmaf(x, x, a2);
    }

  float best = -INFINITY;
  int besti = 0;

  for (int c0 = 0; c0 < n; c0 += kMsCent) {
    const int nc = (n - c0) < kMsCent ? (n - c0) : kMsCent;
    const int nt = (nc + 31) >> 5;
    __syncthreads();  // every wave finished the previous chunk's epilogue (reads b2s)
    if (euclidean) {  // |b|^2 of this chunk's centroids, ascending-k fma chain
      const int c = c0 + threadIdx.x;
      float s = 0.f;
      if (threadIdx.x < nc)
        for (int k = 0; k < d; ++k) {
          const float y = Bb[(int64_t)k * n + c];
          s = fmaf(y, y, s);
        }
      b2s[threadIdx.x] = s;
    }
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int kb = 0; kb < d; kb += kMsKC) {
      __syncthreads();  // previous cs consumers done (also orders b2s)
      for (int e = threadIdx.x; e < kMsKC * kMsCent; e += 256) {
        const int kk = e >> 8, cc = e & 255;
        const int k = kb + kk;
        cs[e] = (k < d && cc < nc) ? Bb[(int64_t)k * n + c0 + cc] : 0.f;
      }
      __syncthreads();
      const int kend = (d - kb) < kMsKC ? (d - kb) : kMsKC;
      for (int kk = 0; kk < kend; kk += 2) {
        const int k = kb + kk + half;
        const float x = (iv && k < d) ? Ab[(int64_t)k * m + i] : 0.f;  // B operand [k][col=point]
        const float* crow = cs + (kk + half) * kMsCent + l31;          // A operand [row=centroid][k]
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(crow[t * 32], x, acc[t], 0, 0, 0);
      }
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
    vals[(int64_t)b * m + i] = best;
    inds[(int64_t)b * m + i] = besti;
  }
}

// ---- update --------------------------------------------------------------------------------
constexpr int kCcPoints = 4096;  // points per block
constexpr int kCcDT = 16;        // dimensions per block

// grid (ceil(n/kCcPoints), ceil(d/kCcDT), l); LDS: [kCcDT][k] sums + [k] counts
__global__ __launch_bounds__(256) void centroid_accum_kernel(const float* __restrict__ data,
                                                             const int64_t* __restrict__ labels,
                                                             float* __restrict__ sums,
                                                             float* __restrict__ counts, int d,
                                                             int64_t n, int k) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* ssum = sh;               // [kCcDT][k]
  float* scnt = sh + kCcDT * k;   // [k]
  const int b = blockIdx.z;
  const int e0 = blockIdx.y * kCcDT;
  const int ne = (d - e0) < kCcDT ? (d - e0) : kCcDT;
  for (int t = threadIdx.x; t < (kCcDT + 1) * k; t += 256) sh[t] = 0.f;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * kCcPoints;
  const int64_t i1 = (i0 + kCcPoints) < n ? (i0 + kCcPoints) : n;
  const bool count_here = (blockIdx.y == 0);
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const int64_t lab = labels[(int64_t)b * n + i];
    if (lab < 0 || lab >= k) continue;
    if (count_here) atomicAdd(&scnt[lab], 1.0f);
    for (int e = 0; e < ne; ++e)
      atomicAdd(&ssum[e * k + lab], data[((int64_t)b * d + e0 + e) * n + i]);
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
  hipLaunchKernelGGL(max_sim_kernel, dim3((m + 127) / 128, l), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), A, B, vals, inds, d, m, n,
                     metric == TPQ_METRIC_NEG_SQ_L2 ? 1 : 0);
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
  if (lds > 160 * 1024) {
    set_error("compute_centroids: k=%d needs %zu bytes of LDS (> 160 KiB)", k, lds);
    return TPQ_ERR_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = check_hip(hipMemsetAsync(workspace, 0, need, st), "compute_centroids memset");
  if (rc) return rc;
  float* sums = reinterpret_cast<float*>(workspace);
  float* counts = sums + (size_t)l * d * k;
  if (n > 0) {
    rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(centroid_accum_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                   "centroid_accum_kernel attr");
    if (rc) return rc;
    hipLaunchKernelGGL(centroid_accum_kernel,
                       dim3((unsigned)((n + kCcPoints - 1) / kCcPoints), (d + kCcDT - 1) / kCcDT, l),
                       dim3(256), lds, st, data, labels, sums, counts, d, n, k);
    TPQ_LAUNCH_CHECK("centroid_accum_kernel");
  }
  const int64_t total = (int64_t)l * d * k;
  hipLaunchKernelGGL(centroid_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st, sums, counts, centroids, d, k, total);
  TPQ_LAUNCH_CHECK("centroid_finalize_kernel");
  return TPQ_OK;
}
