// Interface between tpq_ivfpq_coarse_probe (select.hip) and the fp16 selection pass of lloyd.hip: the coarse step of
// search() at many cells (IVF4096 / IVF16384 of the reference's benchmark grid), where the fp32-MFMA similarity GEMM
// (75 TF/s) was 40-85 % of a search.  The reference offers a reduced-precision coarse GEMM behind use_tensor_core /
// fp16_scale_mode (torchpq/metric.py:47-73, index/IVFPQIndex.py:98-125) and accepts its errors; here the fp16 pass only
// SELECTS: every cell that can still be among the query's n_probe best -- fast value within twice a rigorous error
// bound of the n_probe-th best fast value -- gets the fp32 kernel's own value (the same ascending-k fma chain), and the
// result (cells, order, similarities) is coarse_sims_kernel's, bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace tpq {
struct ProbeFastBuffers {
  const _Float16* sims;  // [nq][n_cells] fast values f' = 2 a'.c' - |c'|^2 (centred, scaled: per query a monotone image of
                         // the similarity), stored as fp16 of f' x qscale[q]
  const float* gmax;     // [nq][n_groups] maxima of the unrounded f' over groups of 2^gshift cells (fp32, not scaled)
  const float* band;     // [nq] 2 delta' x qscale: the candidate band in STORED units before the rounding of the stored
                         // values (which the select kernel adds); +inf: the query is evaluated exactly
  const float* qscale;   // [nq] power of two
  const float* xt;       // [nq][xt_stride] the queries as rows (fp32, as given)
  const float* q2;       // [nq] |x|^2 as the exact kernels sum it (fma chain over ascending k)
  int xt_stride;         // multiple of 4
  const float* ct;    // [n_cells][d] the centroids as rows
  const float* c2;    // [n_cells] |C|^2, ascending-k fma chain
  int n_groups;
  int gshift;            // log2 of the cells per group (5 or 7)
};
int lloyd_probe_supported(int d, int nq, int n_cells);
// groups of cells whose maxima the fast pass keeps (32 cells up to 8 192, 64 up to 16 384, 128 beyond)
int lloyd_probe_groups(int n_cells);
size_t lloyd_probe_workspace_bytes(int d, int nq, int n_cells);
// the part that depends on the centroids alone (mean, scale, fp16 fragments, row copies, |C|^2): once per codebook
size_t lloyd_probe_prepared_bytes(int d, int n_cells);
int lloyd_probe_prepare(const float* centroids, int d, int n_cells, char* prepared, hipStream_t st);
// prepared == nullptr: prepared into the workspace for this call
int lloyd_probe_sims(const float* query, const float* centroids, const void* prepared, int d, int nq, int n_cells,
                     char* ws, ProbeFastBuffers* out, hipStream_t st);
}  // namespace tpq
