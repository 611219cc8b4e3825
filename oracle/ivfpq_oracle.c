/* CPU oracle (C restatement) for the IVFPQ hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain C restatement of the reference's list-scan and k-means-assign
 * arithmetic, used (a) as the checker for larger parity cases that the numpy
 * oracle is too slow for and (b) as the "port" CPU baseline timed by bench.py.
 * Nothing under torchpq_amd/ links or loads this file.
 *
 * Reference (paths relative to the reference repo root):
 *   - scan:     torchpq/kernels/cuda/ivfpq_topk.cu:822-971, consume_data :662-679
 *   - max_sim:  torchpq/kernels/cuda/max_sim.cu:60-98 (fmaf chains), :152-180
 *   - LUT:      torchpq/codec/PQCodec.py:62-75 -> clustering/MultiKMeans.py:184-209
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float v;
  int64_t a;
} cand_t;

/* order: value descending, address ascending (total order, deterministic) */
static int cand_better(const cand_t *x, const cand_t *y) {
  if (x->v > y->v) return 1;
  if (x->v < y->v) return 0;
  return x->a < y->a;
}

/* binary min-heap on "better" (root = worst kept candidate) */
static void heap_sift_down(cand_t *h, int n, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, w = i;
    if (l < n && cand_better(&h[w], &h[l])) w = l;
    if (r < n && cand_better(&h[w], &h[r])) w = r;
    if (w == i) return;
    cand_t t = h[i];
    h[i] = h[w];
    h[w] = t;
    i = w;
  }
}

static int cand_cmp_desc(const void *pa, const void *pb) {
  const cand_t *x = (const cand_t *)pa, *y = (const cand_t *)pb;
  if (cand_better(x, y)) return -1;
  if (cand_better(y, x)) return 1;
  return 0;
}

/* storage: u8 [m/4][n_slots][4]; lut: f32 [m][nq][256]; is_empty: u8[n_slots] or NULL;
 * cell_start/cell_size: i64 [nq][max_nprobe]; n_probe_list: i64 [nq];
 * out_vals f32 [nq][k] (descending, -inf padded), out_addr i64 [nq][k] (-1 padded).
 * Returns total number of scanned (non-tombstoned) slots. */
int64_t oracle_scan_topk(const uint8_t *storage, const float *lut, const uint8_t *is_empty,
                         const int64_t *cell_start, const int64_t *cell_size,
                         const int64_t *n_probe_list, float *out_vals, int64_t *out_addr,
                         int64_t n_slots, int nq, int max_nprobe, int m, int k, int n_threads) {
  int64_t scanned_total = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads) reduction(+ : scanned_total)
  for (int q = 0; q < nq; q++) {
    cand_t *heap = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
    int hn = 0;
    int np = (int)n_probe_list[q];
    if (np > max_nprobe) np = max_nprobe;
    int64_t prev_start = -1;
    int have_prev = 0;
    for (int p = 0; p < np; p++) {
      int64_t st = cell_start[(int64_t)q * max_nprobe + p];
      int64_t sz = cell_size[(int64_t)q * max_nprobe + p];
      /* ivfpq_topk.cu:864-866: a cell whose start equals the previous one is skipped */
      if (have_prev && st == prev_start) continue;
      prev_start = st;
      have_prev = 1;
      for (int64_t s = st; s < st + sz; s++) {
        if (is_empty && is_empty[s]) continue; /* :883-884 */
        float v = 0.f;
        for (int j = 0; j < m; j++) { /* consume_data :662-679, ascending j */
          uint8_t c = storage[((int64_t)(j >> 2) * n_slots + s) * 4 + (j & 3)];
          v += lut[((int64_t)j * nq + q) * 256 + c];
        }
        scanned_total++;
        cand_t cnd = {v, s};
        if (hn < k) {
          heap[hn++] = cnd;
          if (hn == k)
            for (int i = k / 2 - 1; i >= 0; i--) heap_sift_down(heap, k, i);
        } else if (cand_better(&cnd, &heap[0])) {
          heap[0] = cnd;
          heap_sift_down(heap, k, 0);
        }
      }
    }
    qsort(heap, (size_t)hn, sizeof(cand_t), cand_cmp_desc);
    for (int i = 0; i < k; i++) {
      if (i < hn) {
        out_vals[(int64_t)q * k + i] = heap[i].v;
        out_addr[(int64_t)q * k + i] = heap[i].a;
      } else {
        out_vals[(int64_t)q * k + i] = -INFINITY;
        out_addr[(int64_t)q * k + i] = -1;
      }
    }
    free(heap);
  }
  return scanned_total;
}

/* residual-PQ scan: value = base_sims[q][p]; then += (part1[q][j][c] + part2[cell][j][c]) ascending j
 * (ivfpq_topk.cu:1039-1208, LUT build :522-560) or += full[q][p][j][c] when `full` != NULL (:973-1037) */
int64_t oracle_scan_topk_residual(const uint8_t *storage, const float *part1, const float *part2,
                                  const float *full, const int64_t *cells, const float *base_sims,
                                  const uint8_t *is_empty, const int64_t *cell_start,
                                  const int64_t *cell_size, const int64_t *n_probe_list,
                                  float *out_vals, int64_t *out_addr, int64_t n_slots, int nq,
                                  int max_nprobe, int m, int k, int n_threads) {
  int64_t scanned_total = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads) reduction(+ : scanned_total)
  for (int q = 0; q < nq; q++) {
    cand_t *heap = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
    float *lut = (float *)malloc(sizeof(float) * (size_t)m * 256);
    int hn = 0;
    int np = (int)n_probe_list[q];
    if (np > max_nprobe) np = max_nprobe;
    int64_t prev_start = -1;
    int have_prev = 0;
    for (int p = 0; p < np; p++) {
      int64_t st = cell_start[(int64_t)q * max_nprobe + p];
      int64_t sz = cell_size[(int64_t)q * max_nprobe + p];
      if (have_prev && st == prev_start) continue;
      prev_start = st;
      have_prev = 1;
      if (sz <= 0) continue;
      if (full) {
        memcpy(lut, full + ((int64_t)q * max_nprobe + p) * m * 256, sizeof(float) * (size_t)m * 256);
      } else {
        const float *p1 = part1 + (int64_t)q * m * 256;
        const float *p2 = part2 + cells[(int64_t)q * max_nprobe + p] * (int64_t)m * 256;
        for (int e = 0; e < m * 256; e++) lut[e] = p1[e] + p2[e];
      }
      const float base = base_sims[(int64_t)q * max_nprobe + p];
      for (int64_t s = st; s < st + sz; s++) {
        if (is_empty && is_empty[s]) continue;
        float v = base;
        for (int j = 0; j < m; j++) {
          uint8_t c = storage[((int64_t)(j >> 2) * n_slots + s) * 4 + (j & 3)];
          v += lut[j * 256 + c];
        }
        scanned_total++;
        cand_t cnd = {v, s};
        if (hn < k) {
          heap[hn++] = cnd;
          if (hn == k)
            for (int i = k / 2 - 1; i >= 0; i--) heap_sift_down(heap, k, i);
        } else if (cand_better(&cnd, &heap[0])) {
          heap[0] = cnd;
          heap_sift_down(heap, k, 0);
        }
      }
    }
    qsort(heap, (size_t)hn, sizeof(cand_t), cand_cmp_desc);
    for (int i = 0; i < k; i++) {
      if (i < hn) {
        out_vals[(int64_t)q * k + i] = heap[i].v;
        out_addr[(int64_t)q * k + i] = heap[i].a;
      } else {
        out_vals[(int64_t)q * k + i] = -INFINITY;
        out_addr[(int64_t)q * k + i] = -1;
      }
    }
    free(heap);
    free(lut);
  }
  return scanned_total;
}

/* LUT[j][q][c] = 2 q_j.c - |q_j|^2 - |c|^2 (euclidean) or q_j.c (inner);
 * dots and norms are ascending-k fmaf chains (what the fp32 MFMA path computes).
 * query f32 [m*ds][nq], codebook f32 [m][ds][256] -> lut f32 [m][nq][256] */
void oracle_adc_lut(const float *query, const float *codebook, float *lut, int m, int ds, int nq,
                    int euclidean, int n_threads) {
#pragma omp parallel for collapse(2) num_threads(n_threads)
  for (int j = 0; j < m; j++)
    for (int q = 0; q < nq; q++) {
      float q2 = 0.f;
      for (int e = 0; e < ds; e++) {
        float x = query[(int64_t)(j * ds + e) * nq + q];
        q2 = fmaf(x, x, q2);
      }
      for (int c = 0; c < 256; c++) {
        float dot = 0.f, c2 = 0.f;
        for (int e = 0; e < ds; e++) {
          float x = query[(int64_t)(j * ds + e) * nq + q];
          float y = codebook[((int64_t)j * ds + e) * 256 + c];
          dot = fmaf(x, y, dot);
          c2 = fmaf(y, y, c2);
        }
        float r = dot;
        if (euclidean) {
          r = 2.f * dot;
          r = r - q2;
          r = r - c2;
        }
        lut[((int64_t)j * nq + q) * 256 + c] = r;
      }
    }
}

/* A f32 [l][d][m], B f32 [l][d][n] -> vals f32 [l][m], inds i64 [l][m]
 * mode 0: direct  acc = fmaf(-(a-b), (a-b), acc)   (max_sim.cu:78-98)
 * mode 1: inner   acc = fmaf(a, b, acc)            (max_sim.cu:60-75)
 * mode 2: expanded 2*dot - |a|^2 - |b|^2, ascending-k fmaf chains
 * ties -> smallest index */
void oracle_max_sim(const float *A, const float *B, float *vals, int64_t *inds, int l, int d, int m,
                    int n, int mode, int n_threads) {
  for (int b = 0; b < l; b++) {
    const float *Ab = A + (int64_t)b * d * m;
    const float *Bb = B + (int64_t)b * d * n;
    float *b2 = (float *)malloc(sizeof(float) * (size_t)n);
    for (int c = 0; c < n; c++) {
      float s = 0.f;
      for (int k = 0; k < d; k++) s = fmaf(Bb[(int64_t)k * n + c], Bb[(int64_t)k * n + c], s);
      b2[c] = s;
    }
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int i = 0; i < m; i++) {
      float best = -INFINITY;
      int64_t bi = 0;
      float a2 = 0.f;
      for (int k = 0; k < d; k++) a2 = fmaf(Ab[(int64_t)k * m + i], Ab[(int64_t)k * m + i], a2);
      for (int c = 0; c < n; c++) {
        float acc = 0.f;
        if (mode == 0) {
          for (int k = 0; k < d; k++) {
            float dif = Ab[(int64_t)k * m + i] - Bb[(int64_t)k * n + c];
            acc = fmaf(-dif, dif, acc);
          }
        } else {
          for (int k = 0; k < d; k++) acc = fmaf(Ab[(int64_t)k * m + i], Bb[(int64_t)k * n + c], acc);
          if (mode == 2) {
            acc = 2.f * acc;
            acc = acc - a2;
            acc = acc - b2[c];
          }
        }
        if (acc > best) {
          best = acc;
          bi = c;
        }
      }
      vals[(int64_t)b * m + i] = best;
      inds[(int64_t)b * m + i] = bi;
    }
    free(b2);
  }
}

/* Coarse step, euclidean (torchpq/metric.py:75-98 as the index calls it, index/IVFPQIndex.py:485-494):
 * sims[q][c] = (2 dot - |x_q|^2) - |C_c|^2 with dot, |x|^2, |C|^2 ascending-k fmaf chains from 0 --
 * the arithmetic of the fp32-MFMA coarse kernels (an MFMA is an ascending-k fma chain), so the
 * device sims can be checked bit for bit.
 * x f32 [d][nq], C f32 [d][n_cells] -> sims f32 [nq][n_cells] */
void oracle_coarse_sims(const float *x, const float *C, float *sims, int d, int nq, int n_cells,
                        int n_threads) {
  float *c2 = (float *)malloc(sizeof(float) * (size_t)n_cells);
  for (int c = 0; c < n_cells; c++) {
    float s = 0.f;
    for (int k = 0; k < d; k++) s = fmaf(C[(int64_t)k * n_cells + c], C[(int64_t)k * n_cells + c], s);
    c2[c] = s;
  }
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int q = 0; q < nq; q++) {
    float q2 = 0.f;
    for (int k = 0; k < d; k++) q2 = fmaf(x[(int64_t)k * nq + q], x[(int64_t)k * nq + q], q2);
    for (int c = 0; c < n_cells; c++) {
      float acc = 0.f;
      for (int k = 0; k < d; k++) acc = fmaf(C[(int64_t)k * n_cells + c], x[(int64_t)k * nq + q], acc);
      float v = 2.f * acc;
      v = v - q2;
      v = v - c2[c];
      sims[(int64_t)q * n_cells + c] = v;
    }
  }
  free(c2);
}
