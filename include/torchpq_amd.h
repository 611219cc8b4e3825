/* torchpq_amd.h -- C ABI of libtorchpq_amd.so: the MI355X (gfx950) IVFPQ hot path.
 *
 * Drop-in boundary for DeMoriarty/TorchPQ's kernel-wrapper layer
 * (the torchpq/kernels/ xxxCuda.py wrappers).  Every entry point replaces one CuPy RawKernel
 * wrapper of the reference; the reference file:line each one stands in for is
 * cited at its declaration (paths relative to the reference repository root).
 *
 * Conventions (same contract as the reference wrappers):
 *   - all pointers are DEVICE pointers owned by the caller (torch's caching
 *     allocator in the Python host); the library never allocates, frees or
 *     keeps device memory between calls, holds no state between calls and
 *     reads no environment variable (every entry point is re-entrant; the one
 *     third-party exception: tpq_get_ioa sorts with rocPRIM's radix sort, which
 *     consults ROCPRIM_USE_ATOMIC_BLOCK_ID and caches the device architecture);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the null stream) and never synchronises;
 *   - tensors are dense, row-major, in the reference's layouts;
 *   - return value: TPQ_OK (0) or a negative TPQ_ERR_* code; the message is
 *     available from tpq_last_error() (thread-local).
 */
#ifndef TORCHPQ_AMD_H_
#define TORCHPQ_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPQ_VERSION 500 /* 0.5.0 */

#define TPQ_OK 0
#define TPQ_ERR_INVALID_ARGUMENT (-1)
#define TPQ_ERR_HIP (-2)
#define TPQ_ERR_WORKSPACE (-3)
#define TPQ_ERR_UNSUPPORTED (-4)

/* similarity metrics (values are "larger = closer", as in the reference) */
#define TPQ_METRIC_NEG_SQ_L2 0 /* "euclidean": -|a-b|^2 */
#define TPQ_METRIC_INNER 1     /* "cosine"/"inner": a.b (inputs pre-normalised by the host) */

typedef void* tpq_stream_t; /* hipStream_t */

int tpq_version(void);
const char* tpq_last_error(void);

/* ---------------------------------------------------------------------------
 * a-1 / a-2  IVF list scan + top-k
 * replaces IVFPQTopkCuda.topk      torchpq/kernels/IVFPQTopkCuda.py:81-142
 *          IVFPQTop1Cuda.topk      torchpq/kernels/IVFPQTop1Cuda.py:86-140
 * kernels  ivfpq_topk              torchpq/kernels/cuda/ivfpq_topk.cu:822-971
 *          ivfpq_top1              torchpq/kernels/cuda/ivfpq_top1.cu:385-455
 *
 * codes        u8  [m/4][n_slots][4]   CellContainer._storage (contiguous_size=4)
 * lut          f32 [m][nq][256]        PQCodec.precompute_adc output
 * is_empty     u8  [n_slots] or NULL   1 = free/tombstone (NULL: no slot is skipped)
 * cell_start   i64 [nq][max_nprobe]    _cell_start[cells]
 * cell_size    i64 [nq][max_nprobe]    _cell_size[cells]
 * n_probe_list i64 [nq]                cells actually scanned per query (<= max_nprobe)
 * out_vals     f32 [nq][k]             descending; unfilled = -inf
 * out_addr     i64 [nq][k]             slot addresses; unfilled = -1
 * address2id   i64 [n_slots] or NULL   when given, out_ids (i64 [nq][k]) receives
 * out_ids                               BaseContainer.get_id_by_address(out_addr)
 *                                       (torchpq/container/BaseContainer.py:58-65)
 * n_split      workgroups per query (>=1); >1 needs workspace of
 *              tpq_ivfpq_scan_workspace_bytes(nq, k, n_split, m) (the packed variant always
 *              needs it: per-wave candidate lists are merged by a second kernel)
 *
 * value(slot) = 0.f; for j = 0..m-1 ascending: value += lut[j][q][code_j]  (fp32,
 * the order of consume_data, ivfpq_topk.cu:662-679).  Ordering: value descending,
 * exact ties by ascending address.  1 <= k <= 1024, m % 4 == 0, m <= 156.
 * ------------------------------------------------------------------------- */
size_t tpq_ivfpq_scan_workspace_bytes(int nq, int k, int n_split, int m);

int tpq_ivfpq_scan_topk(const uint8_t* codes, const float* lut, const uint8_t* is_empty,
                        const int64_t* cell_start, const int64_t* cell_size,
                        const int64_t* n_probe_list, float* out_vals, int64_t* out_addr,
                        const int64_t* address2id, int64_t* out_ids, int64_t n_slots, int nq,
                        int max_nprobe, int m, int k, int n_split, void* workspace,
                        size_t workspace_bytes, tpq_stream_t stream);

/* MI355X scan layout ("packed"): same bytes as `codes`, permuted per slot so that
 * the 32 lanes of a half-wave always hit 32 distinct LDS banks.
 *   packed u8 [m/W][n_slots][W], W = 16 if m%16==0, else 8 if m%8==0, else 4.
 * tpq_ivfpq_pack_codes (re)builds slots [slot_begin, slot_end) of `packed` from `codes`.
 * tpq_ivfpq_scan_topk_packed has the contract of tpq_ivfpq_scan_topk (bit-identical
 * results) but streams `packed`; `codes` is still needed for the exact re-evaluation of
 * the few candidates that pass the threshold filter.
 * Launches: ONE for plain PQ with k <= 248 when the query is not split (n_split == 1) or the caller
 * passes tickets (the *_tickets entry points) -- the scan workgroups merge their lists, a query split over
 * n_split workgroups is written by the last of them to finish (a ticket per query), and a candidate
 * band that overflowed is redone exactly by that workgroup -- otherwise three (scan, merge, the
 * exact redo of flagged queries).
 * tickets  i32 [nq] (tpq_ivfpq_scan_tickets_bytes(nq) bytes), owned by the CALLER like every other
 *          buffer: all zero before the first call that uses them; every call leaves them all zero
 *          again, so one zeroing serves any number of calls.  Calls that share a ticket buffer must
 *          be ordered (same stream, or otherwise serialised): one buffer per stream in flight.  After
 *          a failed launch or a device fault zero the buffer again.  A graph captured with a ticket
 *          buffer replays with it: it must outlive the graph.  tickets == NULL (and the entry points
 *          without the argument): split queries take the three launches.
 * slots_hint  expected number of slots a query scans (n_probe x mean cell size; 0 = unknown).  Performance
 *          only: long cells let the kernel keep shorter candidate lists per wave (results are the same
 *          either way; a wrong hint costs an exact in-kernel redo of the queries it misleads). */
int tpq_ivfpq_pack_codes(const uint8_t* codes, uint8_t* packed, int64_t n_slots, int m,
                         int64_t slot_begin, int64_t slot_end, tpq_stream_t stream);

int tpq_ivfpq_scan_topk_packed(const uint8_t* packed, const uint8_t* codes, const float* lut,
                               const uint8_t* is_empty, const int64_t* cell_start,
                               const int64_t* cell_size, const int64_t* n_probe_list,
                               float* out_vals, int64_t* out_addr, const int64_t* address2id,
                               int64_t* out_ids, int64_t n_slots, int nq, int max_nprobe, int m,
                               int k, int n_split, void* workspace, size_t workspace_bytes,
                               tpq_stream_t stream);
size_t tpq_ivfpq_scan_tickets_bytes(int nq);

/* Diagnostics: which kernels a scan call with these arguments runs -- a host-side function that launches nothing and
 * applies exactly the rules of the entry points above (tests assert the route a shape takes; bench.py names the timed
 * kernel from it).  `has_lut`: a materialised table is passed (tpq_ivfpq_scan_topk[_packed]) rather than query +
 * codebook (tpq_ivfpq_search_fused); `has_packed`: the scan-layout copy is passed; `residual`: the residual entry point.
 * Returns one of TPQ_SCAN_ROUTE_*, or -1 for arguments the entry points reject. */
#define TPQ_SCAN_ROUTE_REF 0            /* scan_ref_kernel / scan_residual_kernel: the reference layout, exact */
#define TPQ_SCAN_ROUTE_ONE_LAUNCH 1     /* scan_packed_kernel<.., RM > 0>: the scan workgroups finish their queries */
#define TPQ_SCAN_ROUTE_LISTS 2          /* scan_packed_kernel + scan_merge_refine_kernel + the flag-gated exact redo */
#define TPQ_SCAN_ROUTE_POOLS 3          /* large k: scan_packed_kernel<.., RM < 0> + scan_pool_merge_kernel + redo */
#define TPQ_SCAN_ROUTE_DUMP_F32 8       /* large batches, m = 8 / 16 / 32: fp32 table, scan_finish_exact_kernel + redo */
#define TPQ_SCAN_ROUTE_DUMP_SEL16 16    /* large batches, m = 64: 16-bit table, four-wave workgroups + finish + redo */
#define TPQ_SCAN_ROUTE_DUMP_SEL16_W8 17 /* ... eight-wave workgroups (k in (248, 504] on long cells) */
int tpq_ivfpq_scan_route(int nq, int k, int n_split, int m, int ds, int max_nprobe, int64_t slots_hint, int has_lut,
                         int has_packed, int has_tickets, int residual);
int tpq_ivfpq_scan_topk_packed_tickets(const uint8_t* packed, const uint8_t* codes, const float* lut,
                                       const uint8_t* is_empty, const int64_t* cell_start,
                                       const int64_t* cell_size, const int64_t* n_probe_list,
                                       float* out_vals, int64_t* out_addr, const int64_t* address2id,
                                       int64_t* out_ids, int64_t n_slots, int nq, int max_nprobe,
                                       int m, int k, int n_split, void* workspace,
                                       size_t workspace_bytes, int32_t* tickets, int64_t slots_hint,
                                       tpq_stream_t stream);

/* Fused a-3 + a-1: the scan workgroup builds its query's LUT itself from the query and the PQ
 * codebook (bit-identical entries to tpq_adc_lut: same fma chains), so the [m][nq][256] table is
 * never written to or read from HBM.  This is what IVFPQIndex.search_cells
 * (torchpq/index/IVFPQIndex.py:452-461: precompute_adc + IVFPQTopk.topk) collapses to.
 * query f32 [m*ds][nq], codebook f32 [m][ds][256]; `packed` may be NULL (reference-layout
 * kernel); everything else as tpq_ivfpq_scan_topk[_packed].
 * Large batches (nq >= 1024 with n_split == 1, m == 64, ds <= 2, k <= 504, `packed` given): three launches --
 * the scan workgroups (four per CU) stream over a 16-bit fixed-point SELECTION table built from query and
 * codebook and end with their lists of fast values; a finish kernel, one wave per query, evaluates every
 * candidate within the table's rigorous error band exactly (the reference's arithmetic and order, from the
 * codebook) and writes the result; the exact kernel redoes flagged queries (normally none).  Same results,
 * bit for bit.  The library deals the batch's last round of workgroups in parts of a query;
 * nothing of this shows at the boundary beyond the workspace size, which
 * tpq_ivfpq_scan_workspace_bytes(nq, k, n_split, 64) already covers.
 * Short codes (m == 8, 16, 32; k <= 248; m * ds <= 128 here, any table through tpq_ivfpq_scan_topk_packed) take the
 * same three launches over the fp32 table their workgroups stream over anyway: the scan workgroup ends after its
 * last tile, the finish kernel takes the exact entries from the codebook (this entry point) or from the caller's
 * table (tpq_ivfpq_scan_topk_packed).  tpq_ivfpq_scan_route tells which route a shape takes. */
int tpq_ivfpq_search_fused(const uint8_t* packed, const uint8_t* codes, const float* query,
                           const float* codebook, int ds, int metric, const uint8_t* is_empty,
                           const int64_t* cell_start, const int64_t* cell_size,
                           const int64_t* n_probe_list, float* out_vals, int64_t* out_addr,
                           const int64_t* address2id, int64_t* out_ids, int64_t n_slots, int nq,
                           int max_nprobe, int m, int k, int n_split, void* workspace,
                           size_t workspace_bytes, tpq_stream_t stream);
int tpq_ivfpq_search_fused_tickets(const uint8_t* packed, const uint8_t* codes, const float* query,
                                   const float* codebook, int ds, int metric, const uint8_t* is_empty,
                                   const int64_t* cell_start, const int64_t* cell_size,
                                   const int64_t* n_probe_list, float* out_vals, int64_t* out_addr,
                                   const int64_t* address2id, int64_t* out_ids, int64_t n_slots,
                                   int nq, int max_nprobe, int m, int k, int n_split, void* workspace,
                                   size_t workspace_bytes, int32_t* tickets, int64_t slots_hint,
                                   tpq_stream_t stream);

/* ---------------------------------------------------------------------------
 * SURVEY 8(f)-3  residual-PQ list scan (pq_use_residual=True)
 * replaces IVFPQTopkCuda.topk_residual_precomputed  torchpq/kernels/IVFPQTopkCuda.py:212-283
 *          (kernel ivfpq_topk_residual_precomputed, torchpq/kernels/cuda/ivfpq_topk.cu:1039-1208)
 *      and IVFPQTopkCuda.topk_residual              torchpq/kernels/IVFPQTopkCuda.py:144-210
 *          (kernel ivfpq_topk_residual, ivfpq_topk.cu:973-1037)
 * value(slot of probe p) = base_sims[q][p]; then += LUT_p[j][code_j] for j ascending, with
 *   LUT_p = part1[q] + part2[cells[q][p]]  (part1 f32 [nq][m][256], part2 f32 [n_cells][m][256])
 *   or, when full_lut != NULL, LUT_p = full_lut[q][p]  (f32 [nq][max_nprobe][m][256]).
 * Other arguments and the output contract are those of tpq_ivfpq_scan_topk.
 * tpq_residual_part1: part1[q][j][c] = 2 * q_j . r_jc (index/IVFPQIndex.py:366-379).
 * ------------------------------------------------------------------------- */
int tpq_ivfpq_scan_topk_residual(const uint8_t* codes, const float* part1, const float* part2,
                                 const float* full_lut, const int64_t* cells,
                                 const float* base_sims, const uint8_t* is_empty,
                                 const int64_t* cell_start, const int64_t* cell_size,
                                 const int64_t* n_probe_list, float* out_vals, int64_t* out_addr,
                                 const int64_t* address2id, int64_t* out_ids, int64_t n_slots,
                                 int nq, int max_nprobe, int m, int k, tpq_stream_t stream);
int tpq_residual_part1(const float* query, const float* codebook, float* part1, int m, int ds,
                       int nq, tpq_stream_t stream);

/* Residual scan at the speed of the plain scan (same reference kernel, ivfpq_topk.cu:1039-1208,
 * same results bit for bit).  Only part1[q] lives in LDS (staged once per query, or built in the
 * workgroup from query [m*ds][nq] + codebook [m][ds][256] when part1 == NULL); the cell-dependent
 * half of the selection value is a per-slot constant:
 *   tpq_ivfpq_residual_slot_terms: slot_term f32 [n_slots] = sum_j part2[cell(s)][j][code_j(s)]
 *                                  cell_bound f32 [n_cells] = sum_j max_c |part2[cell][j][c]|
 * (derived from CellContainer._storage + part2; recompute after add/remove/expand).
 * Survivors are re-evaluated exactly (base_sims + fl(part1 + part2) ascending j); queries whose
 * candidate band overflows, or that list a cell twice, are redone by the exact kernel.
 * workspace: tpq_ivfpq_scan_workspace_bytes(nq, k, n_split, m).  m: any value of TPQ_PACKED_M_LIST. */
int tpq_ivfpq_residual_slot_terms(const uint8_t* codes, const float* part2,
                                  const int64_t* cell_start, const int64_t* cell_size,
                                  float* slot_term, float* cell_bound, int64_t n_slots,
                                  int n_cells, int m, tpq_stream_t stream);
int tpq_ivfpq_scan_topk_residual_packed(
    const uint8_t* packed, const uint8_t* codes, const float* part1, const float* query,
    const float* codebook, int ds, const float* part2, const float* slot_term,
    const float* cell_bound, const int64_t* cells, const float* base_sims, const uint8_t* is_empty,
    const int64_t* cell_start, const int64_t* cell_size, const int64_t* n_probe_list,
    float* out_vals, int64_t* out_addr, const int64_t* address2id, int64_t* out_ids,
    int64_t n_slots, int nq, int max_nprobe, int m, int k, int n_split, void* workspace,
    size_t workspace_bytes, tpq_stream_t stream);

/* ---------------------------------------------------------------------------
 * a-3  ADC look-up table
 * replaces PQCodec.precompute_adc  torchpq/codec/PQCodec.py:62-75
 *          (-> MultiKMeans.sim/euc_sim, torchpq/clustering/MultiKMeans.py:184-223)
 * query f32 [m*ds][nq], codebook f32 [m][ds][256] -> lut f32 [m][nq][256]
 * NEG_SQ_L2: lut = 2 q.c - |q|^2 - |c|^2 ; INNER: lut = q.c ; every dot/norm is an
 * ascending-dimension fp32 fma chain (exactly what v_mfma_f32_32x32x2_f32 computes).
 * ------------------------------------------------------------------------- */
int tpq_adc_lut(const float* query, const float* codebook, float* lut, int m, int ds, int nq,
                int metric, tpq_stream_t stream);

/* ---------------------------------------------------------------------------
 * a-4  row-wise top-k (coarse probe select)
 * replaces fn.Topk.__call__        torchpq/fn/Topk.py:43-67
 *          Top1SelectCuda / Top32SelectCuda / TopkSelectCuda
 *          (torchpq/kernels/cuda/top1_select.cu:542, top32_select.cu:484-636,
 *           topk_select.cu:662-805)
 * x f32 [rows][cols] -> vals f32 [rows][k] descending, idx i64 [rows][k];
 * ties: smaller column first.  1 <= k <= min(cols, 1024).
 * ------------------------------------------------------------------------- */
int tpq_topk_select(const float* x, float* vals, int64_t* idx, int rows, int cols, int k,
                    tpq_stream_t stream);

/* a-4 fused: the epilogue of metric.negative_squared_l2_distance (torchpq/metric.py:89-96) applied
 * inside the select: v = (2*dots[r][c] - a2[r]) - b2[c] (the reference's rounding order), then
 * row top-k as above.  dots f32 [rows][cols] = x^T C (library GEMM), a2 [rows] = |x|^2,
 * b2 [cols] = |C|^2. */
int tpq_coarse_select(const float* dots, const float* a2, const float* b2, float* vals, int64_t* idx,
                      int rows, int cols, int k, tpq_stream_t stream);

/* a-4 + a-5 + the list-extent gathers in one call: the whole coarse step of IVFPQIndex.search
 * (torchpq/index/IVFPQIndex.py:486-512 and :425-426) as two launches -- an fp32-MFMA kernel that
 * writes sims = 2 x^T C - |x|^2 - |C|^2 (replaces the cuBLAS GEMM + 3 element-wise passes of
 * metric.negative_squared_l2_distance, torchpq/metric.py:31-98) into `workspace`, and the row
 * select whose epilogue also gathers cell_start/cell_size of the chosen cells and derives
 * n_probe_list (smart probing when smart_temperature > 0, else n_probe for every query).
 * query f32 [d][nq], centroids f32 [d][n_cells], cell_*_tbl i64 [n_cells]
 * -> topk_sims f32 [nq][n_probe] (descending), cells / cell_start / cell_size i64 [nq][n_probe],
 *    n_probe_list i64 [nq].   1 <= n_probe <= min(n_cells, 1024). */
size_t tpq_ivfpq_coarse_probe_workspace_bytes(int nq, int n_cells);
int tpq_ivfpq_coarse_probe(const float* query, const float* centroids,
                           const int64_t* cell_start_tbl, const int64_t* cell_size_tbl,
                           float* topk_sims, int64_t* cells, int64_t* cell_start,
                           int64_t* cell_size, int64_t* n_probe_list, int d, int nq, int n_cells,
                           int n_probe, float smart_temperature, void* workspace,
                           size_t workspace_bytes, tpq_stream_t stream);
/* The same call with the choice of the SELECTING arithmetic left to the caller (the result -- cells, order,
 * similarities -- is the fp32 kernel's, bit for bit, on every route).  The reference's reduced-precision coarse GEMM
 * (use_tensor_core / fp16_scale_mode, torchpq/metric.py:47-73, index/IVFPQIndex.py:98-125) trades accuracy for speed;
 * here the fp16 matrix cores only select: fast similarities of all (query, cell) pairs, a candidate band of twice a
 * rigorous error bound around the n_probe-th best, and the fp32 chain's own value for every candidate (d <= 128,
 * n_cells % 32 == 0; other shapes take the fp32 kernels on every route).
 *   TPQ_PROBE_ROUTE_AUTO  what tpq_ivfpq_coarse_probe does: the fp16 selection from 2 048 cells on (batches of more
 *                         than 256 queries; n_probe <= 112, or up to half the number of cell groups the pass keeps
 *                         maxima of -- 128 probes of 16 384 cells) and from 1 024 cells on for batches of >= 4 096
 *                         queries with n_probe <= 32, the fp32 kernels otherwise
 *   TPQ_PROBE_ROUTE_FP32  the fp32-MFMA similarity kernels
 *   TPQ_PROBE_ROUTE_FP16  the fp16 selection whenever the shape allows it
 * workspace: tpq_ivfpq_coarse_probe_route_workspace_bytes(d, nq, n_cells, route). */
#define TPQ_PROBE_ROUTE_AUTO 0
#define TPQ_PROBE_ROUTE_FP32 1
#define TPQ_PROBE_ROUTE_FP16 2
size_t tpq_ivfpq_coarse_probe_route_workspace_bytes(int d, int nq, int n_cells, int route);
int tpq_ivfpq_coarse_probe_route(const float* query, const float* centroids,
                                 const int64_t* cell_start_tbl, const int64_t* cell_size_tbl,
                                 float* topk_sims, int64_t* cells, int64_t* cell_start,
                                 int64_t* cell_size, int64_t* n_probe_list, int d, int nq, int n_cells,
                                 int n_probe, float smart_temperature, int route, const void* prepared,
                                 void* workspace, size_t workspace_bytes, tpq_stream_t stream);
/* What the fp16 selection needs of the CENTROIDS alone (their mean, the fp16 scale, MFMA fragments, row copies,
 * |C|^2) can be prepared once per codebook into a caller-owned block and handed to every call (`prepared`; NULL: it is
 * prepared into the workspace per call, ~0.1 ms at 16 384 cells).  tpq_ivfpq_coarse_probe_prepared_bytes is 0 for shapes
 * without the fp16 pass.  The block is only read by the probe calls. */
size_t tpq_ivfpq_coarse_probe_prepared_bytes(int d, int n_cells);
int tpq_ivfpq_coarse_probe_prepare(const float* centroids, int d, int n_cells, void* prepared,
                                   size_t prepared_bytes, tpq_stream_t stream);

/* a-5  smart probing          torchpq/index/IVFPQIndex.py:499-512
 * topk_sims f32 [rows][n_probe] -> n_probe_list i64 [rows] in [0, n_probe]
 * p = softmax(-sqrt(|s|)/T); H = -sum(p*log2(p)/log2(n_probe)); out = ceil(H*n_probe) */
int tpq_smart_probing(const float* topk_sims, int64_t* n_probe_list, int rows, int n_probe,
                      float temperature, tpq_stream_t stream);

/* id -> address, linear form (BaseContainer(use_inverse_id_mapping=False))
 * replaces get_address_by_id   torchpq/kernels/cuda/get_address_by_id.cu:8-44
 * (wrapper GetAddressByIdCuda, torchpq/container/BaseContainer.py:79-98): every id against every
 * stored id, O(n_ids x capacity); address[i] = smallest slot with address2id == ids[i], else -1
 * (negative ids are never found: free slots hold -1). */
int tpq_get_address_by_id(const int64_t* address2id, int64_t capacity, const int64_t* ids,
                          int64_t* address, int64_t n_ids, tpq_stream_t stream);

/* a-7  address -> id           torchpq/container/BaseContainer.py:58-65 */
int tpq_get_id_by_address(const int64_t* address2id, int64_t capacity, const int64_t* address,
                          int64_t* ids, int64_t n, tpq_stream_t stream);

/* ---------------------------------------------------------------------------
 * a-8  k-means assign / PQ encode: batched arg-max similarity
 * replaces MaxSimCuda.__call__(A, B, dim=2, mode="tn")
 *          torchpq/kernels/MaxSimCuda.py:184-238,296-340; kernel max_sim_tn
 *          torchpq/kernels/cuda/max_sim.cu:182-309
 * A f32 [l][d][m] (data), B f32 [l][d][n] (centroids)
 *   -> vals f32 [l][m], inds i64 [l][m]  (arg-max over the n centroids)
 * NEG_SQ_L2: sim = 2 a.b - |a|^2 - |b|^2, INNER: sim = a.b, dots = ascending-d fp32 fma
 * chains on the fp32 MFMA.  Ties -> smallest centroid index (the reference's cross-block
 * arg-max is a benign race, max_sim.cu:152-180).
 * ------------------------------------------------------------------------- */
int tpq_max_sim(const float* A, const float* B, float* vals, int64_t* inds, int l, int d, int m,
                int n, int metric, tpq_stream_t stream);

/* a-8 (training path)  the same (max, arg-max) on the bf16 matrix cores with fp32-level accuracy
 * replaces the max_sim call of the Lloyd loop, torchpq/clustering/MultiKMeans.py:415-453
 *          (get_labels :301-333 -> MaxSimCuda, kernel max_sim_tn torchpq/kernels/cuda/max_sim.cu:182-309)
 * Every fp32 operand is split exactly into three bf16 pieces and the six piece products of order
 * <= 4 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16: |error of a.b| <= 2^-23 sum|a_k b_k| from
 * the dropped products plus fp32 accumulation rounding -- the accuracy class of the fp32 fma
 * chain, but not its bits: near-ties (relative gap < ~1e-6) may pick a different index than
 * tpq_max_sim.  Shapes: d <= 64, any n (256 centroids per pass), padded slice 16 ceil(d/16) m
 * floats < 2 GiB (tpq_max_sim_split_supported); otherwise TPQ_ERR_UNSUPPORTED. */
int tpq_max_sim_split_supported(int d, int64_t m, int n);
int tpq_max_sim_split(const float* A, const float* B, float* vals, int64_t* inds, int l, int d, int m,
                      int n, int metric, tpq_stream_t stream);

/* a-8 / a-11 (encode path)  the labels of tpq_max_sim, bit for bit, at bf16-matrix-core speed
 * replaces the max_sim call behind VQCodec.encode / IVFPQIndex.add (coarse assign):
 *          torchpq/index/IVFPQIndex.py:233-256 -> clustering/KMeans.py:440-452 (predict)
 *          -> kernels/MaxSimCuda.py:296-340, kernel max_sim_tn torchpq/kernels/cuda/max_sim.cu:182-309
 * A f32 [d][m] points, B f32 [d][n] centroids (one problem) -> inds i64 [m] = arg-max over the
 * centroids of the oracle's fp32 arithmetic (ascending-k fmaf chains; ties -> smallest index).
 * An error-bounded top-2 selection decides every point whose two best fast values differ by more
 * than twice the bound; the others are re-evaluated exactly on the device.  Euclidean problems with
 * >= 4 096 centroids: the points are prepared per call as tpq_lloyd_prepare does, one fp16 product
 * against the centroids in chunks of 256 decides 90-95 % of them, and the rest get the exact kernel's
 * own value for each of their CANDIDATES (the 2-3 centroids within twice the bound of the best);
 * the others the two-piece bf16 selection with the exact kernel over what it leaves.
 * vals (optional, f32 [m]): the maximum itself -- the FAST value (within the bound, ~1e-5 of the
 * scale) for points decided by the selection, the exact one for re-checked points: good for an
 * inertia, not for bit comparisons.
 * Wide vectors, 128 < d <= 1024 (GIST: 960), both metrics: fp16 selection GEMM-shaped (256 x 256 tiles,
 * both operands through LDS), and the exact step runs on CANDIDATES -- for every undecided point the (2-3)
 * centroids within twice the bound of its best get the exact kernel's own value (same instruction
 * sequence), never all n of them; candidate lists that overflow (degenerate data) fall back to the
 * exact kernel.  Problems below 2^33 multiply-adds go to tpq_max_sim directly (same labels).
 * Shapes: d <= 128: m < 2^31, padded slice 16 ceil(d/16) m floats < 2 GiB; 128 < d <= 1024: m < 2^28,
 * n <= 2^22 (workspace ~1.4 x the points); otherwise TPQ_ERR_UNSUPPORTED (use tpq_max_sim).
 * workspace: tpq_coarse_assign_workspace_bytes(d, m, n).
 * tpq_coarse_assign_route: the same call with the choice between the paths made by the caller instead of
 * the size thresholds above -- TPQ_ASSIGN_ROUTE_AUTO (what tpq_coarse_assign does) or
 * TPQ_ASSIGN_ROUTE_CASCADE (the fp16 cascade for every shape it supports, however small: the parity tests
 * drive it over ragged and degenerate shapes this way).  Labels are the same on every route.
 * workspace: tpq_coarse_assign_route_workspace_bytes(d, m, n, route). */
#define TPQ_ASSIGN_ROUTE_AUTO 0
#define TPQ_ASSIGN_ROUTE_CASCADE 1
int tpq_coarse_assign_supported(int d, int64_t m, int n);
size_t tpq_coarse_assign_workspace_bytes(int d, int64_t m, int n);
/* diagnostics: byte offset inside the workspace of the int32 count of points the last call
 * re-checked exactly (d > 128: points that got an exact step on their candidates) */
size_t tpq_coarse_assign_count_offset(int d, int64_t m, int n);
int tpq_coarse_assign(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n,
                      int metric, void* workspace, size_t workspace_bytes, tpq_stream_t stream);
size_t tpq_coarse_assign_route_workspace_bytes(int d, int64_t m, int n, int route);
int tpq_coarse_assign_route(const float* A, const float* B, float* vals, int64_t* inds, int d, int64_t m, int n,
                            int metric, int route, void* workspace, size_t workspace_bytes, tpq_stream_t stream);

/* a-8 (training path, batched)  the labels of tpq_max_sim, bit for bit, for l codebook-sized problems
 * replaces the max_sim call of the Lloyd loop, torchpq/clustering/MultiKMeans.py:415-453
 * A f32 [l][d][m], B f32 [l][d][n], n <= 256, d <= 64 -> inds i64 [l][m] (exact) and, optionally, vals
 * f32 [l][m] (the selection's fast maxima, ~1e-5 of the scale; exact for re-checked points).
 * The error-bounded bf16 top-2 selection + exact re-check of tpq_coarse_assign with the centroids
 * resident in LDS and the points streamed (the PQ-codebook training shape, configs[4]).
 * workspace: tpq_max_sim_select_workspace_bytes(l, d, m, n) (dominated by the l x m int32 lists). */
int tpq_max_sim_select_supported(int l, int d, int64_t m, int n);
size_t tpq_max_sim_select_workspace_bytes(int l, int d, int64_t m, int n);
int tpq_max_sim_select(const float* A, const float* B, float* vals, int64_t* inds, int l, int d, int64_t m,
                       int n, int metric, void* workspace, size_t workspace_bytes, tpq_stream_t stream);

/* a-8 + a-9 on PREPARED data: one Lloyd iteration of the PQ-codebook training shape
 * replaces, per iteration, the get_labels -> compute_centroids pair of the reference's driver
 *          torchpq/clustering/MultiKMeans.py:415-453 (max_sim_tn, kernels/cuda/max_sim.cu:182-309;
 *          compute_centroids, kernels/cuda/compute_centroids.cu:10-86)
 * tpq_lloyd_prepare (once per fit: the data never changes, only the centroids do):
 *   data f32 [l][d][m], centroids0 f32 [l][d][n] (the initial centroids; their mean is the centring
 *   vector) -> `prepared`: every point centred, scaled by a power of two per sub-problem and split
 *   into two fp16 pieces in MFMA-fragment order (4 bytes per element, as the fp32 original), plus
 *   |a'|^2, |x|^2 per point, the centring vector, the scale and a not-finite / out-of-range flag per
 *   sub-problem.  prepared_bytes >= tpq_lloyd_prepared_bytes(l, d, m) (~ the size of data).
 * tpq_lloyd_step: centroids f32 [l][d][n] -> inds i64 [l][m] = the labels of tpq_max_sim, BIT FOR BIT
 *   (error-bounded top-2 selection on the fp16 matrix cores + exact fp32 re-check of the ambiguous
 *   points on `data`), vals (optional) f32 [l][m] = the maxima (fast values, ~1e-6 of the scale; exact
 *   for re-checked points), new_centroids (optional) f32 [l][d][n] = tpq_compute_centroids of those
 *   labels (empty cluster -> 0).  A sub-problem whose data or centroids are not finite / leave the
 *   fp16 range is handled entirely by the exact path (same results, slower).
 * Shapes: d <= 64, n <= 256, slices below 2 GiB (tpq_lloyd_supported); euclidean only. */
int tpq_lloyd_supported(int l, int d, int64_t m, int n);
size_t tpq_lloyd_prepared_bytes(int l, int d, int64_t m);
int tpq_lloyd_prepare(const float* data, const float* centroids0, void* prepared, size_t prepared_bytes,
                      int l, int d, int64_t m, int n, tpq_stream_t stream);
size_t tpq_lloyd_step_workspace_bytes(int l, int d, int64_t m, int n);
/* diagnostics: byte offset inside the workspace of the int32 [l] counts of points left undecided by
 * level 1 (the one-product coarse pass) / level 2 (the three-product pass; = re-checked exactly) */
size_t tpq_lloyd_step_count_offset(int l, int d, int64_t m, int n, int level);
int tpq_lloyd_step(const float* data, const void* prepared, const float* centroids, float* new_centroids,
                   float* vals, int64_t* inds, int l, int d, int64_t m, int n, void* workspace,
                   size_t workspace_bytes, tpq_stream_t stream);

/* a-9  k-means update
 * replaces ComputeCentroidsCuda.__call__  torchpq/kernels/ComputeCentroidsCuda.py:43-81
 *          kernel compute_centroids       torchpq/kernels/cuda/compute_centroids.cu:10-86
 * data f32 [l][d][n], labels i64 [l][n] -> centroids f32 [l][d][k]; empty cluster -> 0.
 * workspace: tpq_compute_centroids_workspace_bytes(l, d, k) (zeroed by the call).
 * Non-finite data (documented divergence): for k <= 256, d >= 32 the sums run as one-hot x data on
 * the bf16 matrix cores, where a NaN / Inf coordinate reaches every cluster of its 16-point group
 * through 0 * x (up to 256 centroids turn NaN in that dimension); the scalar kernels used for the
 * other shapes, and the reference, confine it to the point's own cluster.  Finite inputs are
 * unaffected; callers that may hold NaN / Inf must sanitise the data first. */
size_t tpq_compute_centroids_workspace_bytes(int l, int d, int k);
int tpq_compute_centroids(const float* data, const int64_t* labels, float* centroids, int l, int d,
                          int64_t n, int k, void* workspace, size_t workspace_bytes,
                          tpq_stream_t stream);

/* ---------------------------------------------------------------------------
 * a-12 / a-14  container placement
 * tpq_get_ioa           replaces GetIOACuda            torchpq/kernels/cuda/get_ioa.cu:9-47
 *   ioa[i] = number of j < i with labels[j] == labels[i] (input order).
 *   workspace: tpq_get_ioa_workspace_bytes(n) bytes.
 * tpq_get_write_address replaces GetWriteAddressV2Cuda torchpq/kernels/cuda/get_write_address_v2.cu:9-41
 *   the ioa-th empty slot within [cell_start, cell_start+cell_capacity) of the label's cell, -1 if none.
 * tpq_get_cell_by_address replaces GetDivByAddressV2Cuda torchpq/kernels/cuda/get_div_by_address_v2.cu:9-95
 *   cell whose [start, start+capacity) contains the address, else -1 (cells ordered by start).
 * ------------------------------------------------------------------------- */
size_t tpq_get_ioa_workspace_bytes(int64_t n);
int tpq_get_ioa(const int64_t* labels, int64_t* ioa, int64_t n, int64_t n_cells, void* workspace,
                size_t workspace_bytes, tpq_stream_t stream);
int tpq_get_write_address(const uint8_t* is_empty, const int64_t* cell_start,
                          const int64_t* cell_capacity, const int64_t* labels, const int64_t* ioa,
                          int64_t* write_address, int64_t n_slots, int64_t n_labels,
                          tpq_stream_t stream);
int tpq_get_cell_by_address(const int64_t* address, const int64_t* cell_start,
                            const int64_t* cell_capacity, int64_t* cells, int64_t n_address,
                            int64_t n_cells, tpq_stream_t stream);

/* a-6  expand: re-lay the lists out for larger per-cell capacities
 * replaces CellContainer.expand  torchpq/container/CellContainer.py:249-311 (a torch.cat of the whole
 * storage per expanding cell).  Cell c moves from [old_start[c], + old_capacity[c]) to
 * [new_start[c], + new_capacity[c]) (new_capacity >= old_capacity, new_start = exclusive prefix sum
 * of new_capacity); the added tail of every cell is initialised free (codes 0, id -1, is_empty 1).
 * storage u8 [m/4][old_slots][4] -> new_storage u8 [m/4][new_slots][4]; the new buffers need no
 * initialisation by the caller. */
int tpq_grow_cells(const uint8_t* storage, const int64_t* address2id, const uint8_t* is_empty,
                   const int64_t* old_start, const int64_t* old_capacity, const int64_t* new_start,
                   const int64_t* new_capacity, uint8_t* new_storage, int64_t* new_address2id,
                   uint8_t* new_is_empty, int64_t old_slots, int64_t new_slots, int n_cells, int m,
                   tpq_stream_t stream);

/* a-13  PQ decode   replaces PQDecodeCuda  torchpq/kernels/cuda/pq_decode.cu:8-53
 * codebook f32 [m][ds][256], codes u8 [m][n] -> out f32 [m*ds][n] */
int tpq_pq_decode(const float* codebook, const uint8_t* codes, float* out, int m, int ds, int64_t n,
                  tpq_stream_t stream);

/* codes u8 [m][n] scattered into _storage [m/4][n_slots][4] at address[i]
 * (CellContainer.set_data_by_address, torchpq/container/CellContainer.py:213-239);
 * when `packed` is non-NULL the scan-layout copy is updated in the same pass. */
int tpq_scatter_codes(const uint8_t* codes, const int64_t* address, uint8_t* storage,
                      uint8_t* packed, int m, int64_t n, int64_t n_slots, tpq_stream_t stream);

/* Measurement utility (no reference counterpart): streams `bytes` of `src` through 16-byte
 * loads from `n_blocks` workgroups of 256 threads (0 = 8 per CU) and discards them.  bench.py
 * times it on a buffer larger than the 256 MiB Infinity Cache to obtain the box's sustained HBM
 * read rate, the "measured stream peak" that SURVEY 8d asks roofline fractions to be quoted
 * against next to the 8 TB/s spec figure.  `sink_or_null`: optional u32 the kernel may bump. */
int tpq_ubench_stream_read(const void* src, size_t bytes, void* sink_or_null, int n_blocks,
                           tpq_stream_t stream);
/* Measurement utility: the same streaming read with its knobs exposed -- `threads` per workgroup (a multiple of 64,
 * <= 1024), `unroll` 16-byte loads in flight per lane (4, 8, 16), `chunk_bytes` = the size of a contiguous piece dealt to
 * ONE workgroup (a probed cell of the list scan; 0 = the buffer cut evenly over the workgroups), `nontemporal` loads or
 * plain ones.  bench.py sweeps a few settings, the scan's own pattern among them, and quotes the best as the stream peak:
 * a record labelled "hbm" must not exceed it. */
int tpq_ubench_stream_read_ex(const void* src, size_t bytes, void* sink_or_null, int n_blocks, int threads,
                              int unroll, size_t chunk_bytes, int nontemporal, tpq_stream_t stream);
/* Measurement utility: the k-means update's read pattern without its compute -- data f32
 * [l][d][n] read by one-wave blocks, 32 rows x 256 bytes per 64-point tile, `chunks` blocks per
 * (sub-problem, 32-row group) taking tiles round-robin.  Tells a pattern-bound kernel from a
 * compute-bound one (tools/kmeans_microbench.py --rows-read). */
int tpq_ubench_rows_read(const float* src, int l, int d, int64_t n, int chunks, void* sink_or_null,
                         tpq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TORCHPQ_AMD_H_ */
