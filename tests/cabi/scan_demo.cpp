// C-ABI smoke test without Python or torch: plain hipMalloc'ed buffers in, results out.
//
//   hipcc --offload-arch=gfx950 -O2 -I include tests/cabi/scan_demo.cpp \
//         -L torchpq_amd -ltorchpq_amd -Wl,-rpath,$PWD/torchpq_amd -o tests/cabi/scan_demo
//
// Builds a small inverted-list index on the host (reference layout, ragged cells with slack),
// calls tpq_ivfpq_scan_topk and -- after tpq_ivfpq_pack_codes -- tpq_ivfpq_scan_topk_packed through
// include/torchpq_amd.h, and checks both against a scalar restatement of the reference kernel's
// arithmetic (sum_j LUT[j][code_j] in ascending j, ivfpq_topk.cu:662-679; ties by ascending address);
// then tpq_coarse_assign against a scalar restatement of the fp32 assign arithmetic.
// Test infrastructure (run by tests/test_gpu_cabi_demo.py); exit code 0 = bit-exact.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "torchpq_amd.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define TPQ_OK_(x)                                                                 \
  do {                                                                             \
    int rc_ = (x);                                                                 \
    if (rc_ != 0) {                                                                \
      fprintf(stderr, "%s:%d rc=%d: %s\n", __FILE__, __LINE__, rc_, tpq_last_error()); \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return rng_state >> 8;
}

template <class T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main() {
  const int m = 64, nq = 37, n_cells = 40, n_probe = 9, k = 50;
  // ragged cells with slack
  std::vector<int64_t> cell_start(n_cells), cell_size(n_cells);
  int64_t n_slots = 0;
  for (int c = 0; c < n_cells; ++c) {
    cell_start[c] = n_slots;
    cell_size[c] = (rnd() % 10 == 0) ? 0 : 100 + rnd() % 200;
    n_slots += cell_size[c] + rnd() % 7;
  }
  n_slots += 5;
  std::vector<uint8_t> codes((size_t)m * n_slots);          // [m/4][n_slots][4]
  for (auto& b : codes) b = (uint8_t)(rnd() & 255);
  std::vector<float> lut((size_t)m * nq * 256);             // [m][nq][256]
  for (auto& v : lut) v = (float)((int)(rnd() % 20001) - 10000) / 37.0f;
  std::vector<int64_t> q_start((size_t)nq * n_probe), q_size((size_t)nq * n_probe), n_probe_list(nq);
  for (int q = 0; q < nq; ++q) {
    n_probe_list[q] = (q < 5) ? n_probe : (int64_t)(rnd() % (n_probe + 1));
    for (int p = 0; p < n_probe; ++p) {
      const int c = (int)((q * 7 + p * 11) % n_cells);      // distinct cells per query
      q_start[(size_t)q * n_probe + p] = cell_start[c];
      q_size[(size_t)q * n_probe + p] = cell_size[c];
    }
  }

  // scalar restatement: value desc, address asc
  struct Cand { float v; int64_t a; };
  std::vector<float> exp_v((size_t)nq * k, -INFINITY);
  std::vector<int64_t> exp_a((size_t)nq * k, -1);
  for (int q = 0; q < nq; ++q) {
    std::vector<Cand> all;
    for (int p = 0; p < (int)n_probe_list[q]; ++p) {
      const int64_t st = q_start[(size_t)q * n_probe + p], sz = q_size[(size_t)q * n_probe + p];
      for (int64_t s = st; s < st + sz; ++s) {
        float v = 0.f;
        for (int j = 0; j < m; ++j)
          v += lut[((size_t)j * nq + q) * 256 + codes[((size_t)(j >> 2) * n_slots + s) * 4 + (j & 3)]];
        all.push_back({v, s});
      }
    }
    std::sort(all.begin(), all.end(), [](const Cand& x, const Cand& y) {
      return x.v > y.v || (x.v == y.v && x.a < y.a);
    });
    for (int i = 0; i < k && i < (int)all.size(); ++i) {
      exp_v[(size_t)q * k + i] = all[i].v;
      exp_a[(size_t)q * k + i] = all[i].a;
    }
  }

  uint8_t* d_codes = to_device(codes);
  float* d_lut = to_device(lut);
  int64_t* d_start = to_device(q_start);
  int64_t* d_size = to_device(q_size);
  int64_t* d_npl = to_device(n_probe_list);
  if (!d_codes || !d_lut || !d_start || !d_size || !d_npl) return 2;
  uint8_t* d_packed = nullptr;
  float* d_vals = nullptr;
  int64_t* d_addr = nullptr;
  void* d_ws = nullptr;
  HIP_OK(hipMalloc(&d_packed, codes.size()));
  HIP_OK(hipMalloc(&d_vals, (size_t)nq * k * sizeof(float)));
  HIP_OK(hipMalloc(&d_addr, (size_t)nq * k * sizeof(int64_t)));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  std::vector<float> got_v((size_t)nq * k);
  std::vector<int64_t> got_a((size_t)nq * k);
  int failures = 0;
  for (int variant = 0; variant < 4; ++variant) {  // reference layout / scan layout, 1 or 3 splits
    const int n_split = (variant & 1) ? 3 : 1;
    const bool packed = variant >= 2;
    const size_t ws_bytes = tpq_ivfpq_scan_workspace_bytes(nq, k, n_split, m);
    if (d_ws) HIP_OK(hipFree(d_ws));
    HIP_OK(hipMalloc(&d_ws, ws_bytes + 16));
    HIP_OK(hipMemsetAsync(d_vals, 0, (size_t)nq * k * sizeof(float), stream));
    if (packed) {
      TPQ_OK_(tpq_ivfpq_pack_codes(d_codes, d_packed, n_slots, m, 0, n_slots, stream));
      TPQ_OK_(tpq_ivfpq_scan_topk_packed(d_packed, d_codes, d_lut, nullptr, d_start, d_size, d_npl,
                                         d_vals, d_addr, nullptr, nullptr, n_slots, nq, n_probe, m, k,
                                         n_split, d_ws, ws_bytes, stream));
    } else {
      TPQ_OK_(tpq_ivfpq_scan_topk(d_codes, d_lut, nullptr, d_start, d_size, d_npl, d_vals, d_addr,
                                  nullptr, nullptr, n_slots, nq, n_probe, m, k, n_split, d_ws,
                                  ws_bytes, stream));
    }
    HIP_OK(hipMemcpyAsync(got_v.data(), d_vals, got_v.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(got_a.data(), d_addr, got_a.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    int bad = 0;
    for (size_t i = 0; i < got_v.size(); ++i)
      if (got_a[i] != exp_a[i] || !(got_v[i] == exp_v[i])) ++bad;
    printf("%s layout, n_split=%d: %s (%d mismatches of %zu)\n", packed ? "scan" : "reference", n_split,
           bad ? "FAIL" : "bit-exact", bad, got_v.size());
    failures += bad != 0;
  }
  // argument validation comes back as an error code with a message, not a crash
  const int rc = tpq_ivfpq_scan_topk(d_codes, d_lut, nullptr, d_start, d_size, d_npl, d_vals, d_addr,
                                     nullptr, nullptr, n_slots, nq, n_probe, /*m=*/6, k, 1, d_ws, 0, stream);
  printf("bad m -> rc=%d (%s)\n", rc, tpq_last_error());
  failures += rc == 0;
  // ---- tpq_coarse_assign: the labels of the fp32 arithmetic (ascending-k fmaf chains, (2 acc - |a|^2)
  // - |c|^2, ties -> smallest index: oracle_max_sim mode "expanded"), from the bounded bf16 selection
  // + exact re-check; centroids include exact duplicates and near-duplicates
  {
    const int d = 96, np = 3000, nc = 700;
    std::vector<float> A((size_t)d * np), B((size_t)d * nc);
    for (auto& v : A) v = (float)((int)(rnd() % 2001) - 1000) / 64.0f;
    for (int c = 0; c < nc; ++c)
      for (int kk = 0; kk < d; ++kk) {
        const int src = (c % 3 == 0) ? (c / 3) % np : (int)(rnd() % np);  // every third one copies a point
        B[(size_t)kk * nc + c] = A[(size_t)kk * np + src] + ((c % 3 == 0) ? 0.f : (float)((int)(rnd() % 201) - 100) / 512.0f);
      }
    for (int kk = 0; kk < d; ++kk) B[(size_t)kk * nc + 5] = B[(size_t)kk * nc + 0];  // an exact duplicate
    std::vector<int64_t> want(np);
    std::vector<float> b2(nc);
    for (int c = 0; c < nc; ++c) {
      float sq = 0.f;
      for (int kk = 0; kk < d; ++kk) sq = fmaf(B[(size_t)kk * nc + c], B[(size_t)kk * nc + c], sq);
      b2[c] = sq;
    }
    for (int i = 0; i < np; ++i) {
      float a2 = 0.f;
      for (int kk = 0; kk < d; ++kk) a2 = fmaf(A[(size_t)kk * np + i], A[(size_t)kk * np + i], a2);
      float best = -INFINITY;
      int64_t bi = 0;
      for (int c = 0; c < nc; ++c) {
        float acc = 0.f;
        for (int kk = 0; kk < d; ++kk) acc = fmaf(A[(size_t)kk * np + i], B[(size_t)kk * nc + c], acc);
        acc = 2.f * acc;
        acc = acc - a2;
        acc = acc - b2[c];
        if (acc > best) {
          best = acc;
          bi = c;
        }
      }
      want[i] = bi;
    }
    float* dA = to_device(A);
    float* dB = to_device(B);
    int64_t* d_lab = nullptr;
    void* d_cws = nullptr;
    if (!dA || !dB) return 2;
    HIP_OK(hipMalloc(&d_lab, (size_t)np * sizeof(int64_t)));
    const size_t cws = tpq_coarse_assign_workspace_bytes(d, np, nc);
    HIP_OK(hipMalloc(&d_cws, cws + 16));
    TPQ_OK_(tpq_coarse_assign(dA, dB, nullptr, d_lab, d, np, nc, TPQ_METRIC_NEG_SQ_L2, d_cws, cws, stream));
    std::vector<int64_t> lab(np);
    HIP_OK(hipMemcpyAsync(lab.data(), d_lab, lab.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    int bad = 0;
    for (int i = 0; i < np; ++i) bad += lab[i] != want[i];
    printf("coarse assign, %d points x %d centroids x %d: %s (%d mismatches)\n", np, nc, d,
           bad ? "FAIL" : "labels bit-exact", bad);
    failures += bad != 0;
  }
  printf("tpq_version %d\n", tpq_version());
  return failures ? 1 : 0;
}
