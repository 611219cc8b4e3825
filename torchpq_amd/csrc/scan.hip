// IVF list scan + top-k for gfx950 (MI355X).
//
// Replaces ivfpq_topk / ivfpq_top1 (torchpq/kernels/cuda/ivfpq_topk.cu:822-971,
// ivfpq_top1.cu:385-455) and their launchers (torchpq/kernels/IVFPQTopkCuda.py:81-142).
//
// Structure:
//   * one workgroup per (query, split); the query's 256-entry-per-sub-quantizer LUT (m KiB fp32)
//     is staged in LDS; each wave walks its round-robin share of the probed cells in 64-slot
//     tiles, one slot per lane, codes streamed straight from HBM to VGPRs;
//   * value(slot) = sum_j LUT[j][code_j] in fp32, ascending j from 0.f -- bit-identical to
//     consume_data (ivfpq_topk.cu:662-679);
//   * per-wave register top-k (wave_topk.h) + a workgroup-shared admission threshold in LDS;
//     no barrier inside the scan loop;
//   * n_split > 1 splits a query's tiles over several workgroups (small batches must still
//     fill 256 CUs; the reference's grid=(nq,) cannot).
//
// Kernels:
//   scan_ref_kernel      streams CellContainer._storage as is ([m/4][n_slots][4], any m % 4 == 0):
//                        the drop-in at the IVFPQTopkCuda.topk boundary and the exact fallback.
//                        LDS look-ups hit random banks (~3.5 cycles per half-wave access); the 8
//                        waves' lists are tree-merged in the workgroup.
//   scan_residual_kernel residual PQ (per-cell LUT), reference layout, exact.
//   scan_packed_kernel   streams the MI355X scan layout (pack.hip / scan_layout.h): per-slot
//                        XOR-permuted sub-quantizer order so the 32 lanes of a half-wave always
//                        read 32 distinct banks.  The permuted summation order makes its value
//                        a SELECTION key only; each wave re-evaluates its few surviving
//                        candidates exactly (ascending j) at the end of the query and dumps its
//                        list; scan_merge_refine_kernel (one wave per query) merges the lists.
//                        Results are bit-identical to scan_ref_kernel.
#include <atomic>
#include <chrono>

#include "scan_device.h"

namespace tpq {

// part1[q][j][c] = 2 * (q_j . r_jc), dots as ascending-dimension fma chains
// (IVFPQIndex.precomputed_adc_residual_precomputed, index/IVFPQIndex.py:366-379)
__global__ __launch_bounds__(256) void residual_part1_kernel(const float* __restrict__ query,
                                                            const float* __restrict__ codebook,
                                                            float* __restrict__ part1, int m,
                                                            int ds, int nq) {
  const int j = blockIdx.y, q = blockIdx.x, c = threadIdx.x;
  float dot = 0.f;
  for (int e = 0; e < ds; ++e)
    dot = fmaf(query[(int64_t)(j * ds + e) * nq + q], codebook[((int64_t)j * ds + e) * 256 + c], dot);
  part1[((int64_t)q * m + j) * 256 + c] = 2.f * dot;
}


template <int R>
static int launch_ref(ScanArgs a, hipStream_t st) {
  const size_t lds = scan_lds_bytes_ref(a.m, R, a.max_nprobe, fused_floats_of(a));
  int rc = set_lds(scan_ref_kernel<R>, lds, "scan_ref_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(scan_ref_kernel<R>, dim3((unsigned)a.nq * a.n_split), dim3(kScanThreads), lds,
                     st, a);
  TPQ_LAUNCH_CHECK("scan_ref_kernel");
  if (a.n_split > 1) {
    hipLaunchKernelGGL(scan_merge_kernel<R>, dim3(a.nq), dim3(64), 0, st, a);
    TPQ_LAUNCH_CHECK("scan_merge_kernel");
  }
  return TPQ_OK;
}

static int dispatch_ref(const ScanArgs& a, int R, hipStream_t st) {
  switch (R) {
    case 1: return launch_ref<1>(a, st);
    case 2: return launch_ref<2>(a, st);
    case 4: return launch_ref<4>(a, st);
    case 8: return launch_ref<8>(a, st);
    default: return launch_ref<16>(a, st);
  }
}

}  // namespace tpq

using namespace tpq;

extern "C" size_t tpq_ivfpq_scan_workspace_bytes(int nq, int k, int n_split, int m) {
  if (nq <= 0 || k <= 0) return 0;
  if (n_split < 1) n_split = 1;
  int R = list_regs_packed(k) > list_regs(k) ? list_regs_packed(k) : list_regs(k);
  if (R > 16) R = 16;
  // covers both kernels -- and, at m = 64, the dump route's split tail: 4 parts x 4 waves of lists per query
  size_t lists = ws_bytes_for(nq, R, n_split * packed_waves(m));
  if ((m == 64 || m == 32 || m == 16 || m == 8) && dump_finish_regs(k, 1) <= kDumpMaxR) {
    const int rp = list_regs_packed(k);
    const size_t tail = ws_bytes_for(nq, rp < 2 ? rp : 2, 16), whole = ws_bytes_for(nq, rp < 4 ? rp : 4, 4);
    lists = tail > lists ? tail : lists;
    lists = whole > lists ? whole : lists;
  }
  const size_t pools = list_regs_packed(k) >= pool_min_list_regs(m) ? pool_ws_bytes(nq, k, m, n_split * packed_waves(m)) : 0;  // pool mode
  return lists > pools ? lists : pools;
}

static int run_ref(ScanArgs a, void* workspace, size_t workspace_bytes, tpq_stream_t stream);
static int run_packed(ScanArgs a, const ResidualArgs* ra, void* workspace, size_t workspace_bytes,
                      tpq_stream_t stream);
#ifdef TPQ_SCAN_PROFILE
// private instrumentation build: phase timestamps of scan_packed_kernel (tools/scan_phase_profile.py)
static unsigned long long* g_scan_prof = nullptr;
extern "C" void tpq_debug_set_scan_profile(void* buf) { g_scan_prof = (unsigned long long*)buf; }
#endif
// ---- tickets of the fused finish ---------------------------------------------------------------------
// A query split over several workgroups is finished by the last one to arrive: one int32 ticket per query,
// ZERO when the kernel starts and zero again when it ends (the finisher resets it).  The tickets belong to the
// CALLER (tpq_ivfpq_scan_tickets_bytes; zeroed once, then reusable by every later call that is ordered after
// this one): the workspace arrives uninitialised and zeroing it would be a launch of its own, and a buffer
// owned by the library would be state the boundary does not allow.  No tickets: a split query takes the
// three-launch path (scan, merge, flagged redo).
static bool fuse_enabled() {
  static const bool v = [] {
    const char* e = TPQ_AB_ENV("TPQ_SCAN_FUSE");  // (variant builds, A/B: TPQ_SCAN_FUSE=0 runs the three-launch path)
    return !(e && atoi(e) == 0);
  }();
  return v;
}
// the value that marks a raised flag in this call: non-zero, different from call to call (the flags live in
// the caller's uninitialised workspace, which may be the memory an earlier call used) -- from the clock, so
// that the library keeps no counter
// INVARIANT (captured graphs replay with the SAME epoch and the same workspace): every path that raises flags also
// consumes them before the call's last kernel ends -- the fused finisher and scan_ref_kernel / scan_residual_kernel
// reset a flag they act on; scan_pool_merge_kernel and scan_finish_exact_kernel only skip (or raise) and rely on the
// dispatch_ref launch that follows them IN THE SAME CALL.  A call that returns an error between those two launches
// leaves flags raised: harmless for the next call (another epoch), and a capture that saw a failed launch is dead
// anyway.  A stale word that happens to equal the epoch costs one needless exact redo, never a wrong result.
static int fresh_epoch() {
  const uint64_t ns = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  return (int)(((uint32_t)(ns ^ (ns >> 29)) * 0x9e3779b1u) | 1u);
}

// large batches of plain PQ at m = 64, k <= 504: dump mode over the 16-bit table (scan_device.h).
// (variants: TPQ_SCAN_DUMP=0 keeps the one-launch finish / the lists)
// returns 0 (not this route), kDumpSel16 (four-wave workgroups) or kDumpSel16W8 (eight-wave workgroups: k in (248, 504]
// where four waves would need longer lists than eight -- long cells; lists of <= 2 registers, 16 chunks per query)
static int dump_min_queries() {  // (variants: TPQ_SCAN_DUMP_MINQ=n takes batches from n queries on, whatever n_split says)
  const char* e = TPQ_AB_ENV("TPQ_SCAN_DUMP_MINQ");
  return e ? atoi(e) : kDumpMinQueries;
}
static int dump_route(const ScanArgs& a, bool residual, int R) {
  if (residual || R > kDumpMaxR) return 0;
  // (the route deals the queries itself -- dump_tail --: a caller's n_split > 1 belongs to batches below its threshold)
  if (a.n_split != 1 && dump_min_queries() == kDumpMinQueries) return 0;
  if (a.nq < dump_min_queries()) return 0;
  const char* e = TPQ_AB_ENV("TPQ_SCAN_DUMP");
  if (e && atoi(e) == 0) return 0;
  // (the finish kernel takes a query's lists as at most 16 chunks of 64 keys: four waves x RL <= 4, eight x RL <= 2)
  const int rl4 = list_regs_scan(a.k, a.m, a.max_nprobe, a.slots_hint, 4);
  if (a.m == 8 || a.m == 16 || a.m == 32) {
    // Short codes (round 6): the fp32 table their four-wave workgroups stream over anyway; the finish kernel takes the
    // exact entries from the codebook in LDS (fused calls, m * ds <= 128) or from the caller's table.  k <= 248: the
    // pools keep the larger k (TPQ_SCAN_DUMP_SHORT_K=n in variant builds moves the limit for A/B).
    const char* ek = TPQ_AB_ENV("TPQ_SCAN_DUMP_SHORT_K");
    const int kmax = ek ? atoi(ek) : kDumpShortMaxK;
    if (a.k > kmax || rl4 > 4) return 0;
    if (!a.lut && a.m * a.ds > 128) return 0;
    // a caller's table: the finish kernel gathers m entries per survivor from it -- pays only behind long scans (the
    // reference grid at m = 16, IVF4096 cells of 244 slots, scan fraction of 8 TB/s, one-launch finish / this route:
    // 32 probes 0.44 / 0.39, 64 probes 0.49 / 0.50, 128 probes 0.58 / 0.63)
    if (a.lut && a.slots_hint < kDumpLutMinSlots) return 0;
    return kDumpF32;
  }
  // m = 64: the 16-bit table.  The finish kernel recomputes the survivors' table entries from the codebook, held in
  // LDS next to nothing else: fused calls (query + codebook) only, m * ds <= 128
  if (a.lut || a.m != 64 || a.ds > 2) return 0;
  const int rl8 = list_regs_scan(a.k, a.m, a.max_nprobe, a.slots_hint);
  // Lists of 4 registers in four-wave workgroups only where eight waves would need them as well (short cells, k > 256):
  // folding into a 256-entry list is what a large k costs, and four waves see twice the candidates each.
  // Same box, 10 000 queries, ms, four waves / the sorted lists of the three-launch path: k = 300, 244-slot cells x
  // 16 / 32 / 64 probes: 1.48 / 1.98 / 2.92 against 2.23 / 2.83 / 3.91 (k = 500: 1.74 / 2.51 / 3.61 against 2.63 / 3.33
  // / 4.47) -- both at 4 registers; 977-slot cells x 8 / 32 probes, k = 300: 1.91 / 4.37 against 1.69 / 3.45 -- 4
  // registers against 2: those take the eight-wave form.
  if (rl4 <= 2 || (rl4 <= 4 && rl4 <= rl8)) return kDumpSel16;
  if (e && atoi(e) == 1) return 0;  // (variants: TPQ_SCAN_DUMP=1 = four waves or the lists)
  if (a.k > 248 && rl8 <= 2) return kDumpSel16W8;
  return 0;
}

// The batch's last round of workgroups (ScanArgs::unsplit): the chip holds 4 dump-mode workgroups per CU; the
// nq mod slots queries left after the full rounds are dealt in 4 (or 2) parts each when those parts fit one round and
// the finish kernel can take their lists (<= 16 chunks of 64 keys per query) -- 1 250 queries: 1 024 whole + 226 x 4.
static int device_cus() {
  // (per-device cache; relaxed atomics: concurrent first calls from several host threads write the same value)
  static std::atomic<int> cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n <= 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
// workgroup slots of the dump-mode scan kernel on this device: CUs x the workgroups of it a CU holds -- from the occupancy
// query of the kernel's smallest instantiation (the "4 per CU" used to be an assumption)
static int dump_slots(int m, int mode) {
  static std::atomic<int> per_cu[3][4] = {};  // [mode][m class]
  const int mi = mode == kDumpSel16 ? 0 : (mode == kDumpF32 ? 1 : 2), ci = m == 64 ? 0 : (m == 32 ? 1 : (m == 16 ? 2 : 3));
  int n = per_cu[mi][ci].load(std::memory_order_relaxed);
  if (n <= 0) {
    n = 4;
    int q = 0;
    switch (m) {
#define TPQ_CASE_M(M) case M: q = dump_occupancy_##M(mode); break;
      TPQ_PACKED_M_LIST(TPQ_CASE_M)
#undef TPQ_CASE_M
      default: break;
    }
    if (q > 0) n = q;
    per_cu[mi][ci].store(n, std::memory_order_relaxed);
  }
  return n * device_cus();
}
static void dump_tail(int nq, int nw, int RL, int slots, int* unsplit, int* parts) {
  *unsplit = nq;
  *parts = 1;
  if (slots <= 0) return;
  const int whole = (nq / slots) * slots, rest = nq - whole;
  if (rest == 0) return;
  for (int c = 4; c >= 2; c >>= 1) {
    if ((int64_t)rest * c <= slots && c * nw * RL <= 16) {
      *unsplit = whole;
      *parts = c;
      return;
    }
  }
}

static bool has_packed_kernel(int m) {
#define TPQ_IS_M(M) if (m == M) return true;
  TPQ_PACKED_M_LIST(TPQ_IS_M)
#undef TPQ_IS_M
  return false;
}
static int run_residual_ref(ScanArgs a, ResidualArgs ra, hipStream_t st);

extern "C" size_t tpq_ivfpq_scan_tickets_bytes(int nq) { return nq > 0 ? (size_t)nq * sizeof(int32_t) : 0; }

extern "C" int tpq_ivfpq_search_fused_tickets(const uint8_t* packed, const uint8_t* codes,
                                              const float* query, const float* codebook, int ds,
                                              int metric, const uint8_t* is_empty,
                                              const int64_t* cell_start, const int64_t* cell_size,
                                              const int64_t* n_probe_list, float* out_vals,
                                              int64_t* out_addr, const int64_t* address2id,
                                              int64_t* out_ids, int64_t n_slots, int nq, int max_nprobe,
                                              int m, int k, int n_split, void* workspace,
                                              size_t workspace_bytes, int32_t* tickets,
                                              int64_t slots_hint, tpq_stream_t stream) {
  TPQ_REQUIRE(metric == TPQ_METRIC_NEG_SQ_L2 || metric == TPQ_METRIC_INNER,
              "ivfpq_search_fused: bad metric %d", metric);
  TPQ_REQUIRE(ds >= 1 && ds <= 1024, "ivfpq_search_fused: bad sub-vector length %d", ds);
  ScanArgs a{codes, packed, nullptr, query, codebook, ds, metric == TPQ_METRIC_NEG_SQ_L2 ? 1 : 0,
             is_empty, cell_start, cell_size, n_probe_list, out_vals, out_addr, address2id, out_ids,
             nullptr, nullptr, nullptr, nullptr, nullptr, n_slots, nq, max_nprobe, m, k, n_split};
  a.tickets = tickets;
  a.slots_hint = slots_hint;
  if (packed && has_packed_kernel(m)) return run_packed(a, nullptr, workspace, workspace_bytes, stream);
  return run_ref(a, workspace, workspace_bytes, stream);
}

extern "C" int tpq_ivfpq_search_fused(const uint8_t* packed, const uint8_t* codes,
                                      const float* query, const float* codebook, int ds,
                                      int metric, const uint8_t* is_empty,
                                      const int64_t* cell_start, const int64_t* cell_size,
                                      const int64_t* n_probe_list, float* out_vals,
                                      int64_t* out_addr, const int64_t* address2id,
                                      int64_t* out_ids, int64_t n_slots, int nq, int max_nprobe,
                                      int m, int k, int n_split, void* workspace,
                                      size_t workspace_bytes, tpq_stream_t stream) {
  return tpq_ivfpq_search_fused_tickets(packed, codes, query, codebook, ds, metric, is_empty, cell_start,
                                        cell_size, n_probe_list, out_vals, out_addr, address2id, out_ids,
                                        n_slots, nq, max_nprobe, m, k, n_split, workspace, workspace_bytes,
                                        nullptr, 0, stream);
}

extern "C" int tpq_ivfpq_scan_topk(const uint8_t* codes, const float* lut, const uint8_t* is_empty,
                                   const int64_t* cell_start, const int64_t* cell_size,
                                   const int64_t* n_probe_list, float* out_vals, int64_t* out_addr,
                                   const int64_t* address2id, int64_t* out_ids, int64_t n_slots,
                                   int nq, int max_nprobe, int m, int k, int n_split,
                                   void* workspace, size_t workspace_bytes, tpq_stream_t stream) {
  ScanArgs a{codes, nullptr, lut, nullptr, nullptr, 0, 0, is_empty, cell_start, cell_size,
             n_probe_list, out_vals, out_addr, address2id, out_ids, nullptr, nullptr, nullptr,
             nullptr, nullptr, n_slots, nq, max_nprobe, m, k, n_split};
  return run_ref(a, workspace, workspace_bytes, stream);
}

static int run_ref(ScanArgs a, void* workspace, size_t workspace_bytes, tpq_stream_t stream) {
  int rc = validate(a);
  if (rc) return rc;
  if (a.nq == 0) return TPQ_OK;
  const int nq = a.nq, k = a.k, n_split = a.n_split;
  const int R = list_regs(k);
  if (n_split > 1) {
    rc = need_ws(workspace, workspace_bytes, ws_bytes_for(nq, R, n_split), "ivfpq_scan");
    if (rc) return rc;
    fill_ws(a, workspace, R, n_split);
  }
  return dispatch_ref(a, R, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int tpq_ivfpq_scan_topk_packed_tickets(const uint8_t* packed, const uint8_t* codes,
                                                  const float* lut, const uint8_t* is_empty,
                                                  const int64_t* cell_start, const int64_t* cell_size,
                                                  const int64_t* n_probe_list, float* out_vals,
                                                  int64_t* out_addr, const int64_t* address2id,
                                                  int64_t* out_ids, int64_t n_slots, int nq, int max_nprobe,
                                                  int m, int k, int n_split, void* workspace,
                                                  size_t workspace_bytes, int32_t* tickets,
                                                  int64_t slots_hint, tpq_stream_t stream) {
  ScanArgs a{codes, packed, lut, nullptr, nullptr, 0, 0, is_empty, cell_start, cell_size,
             n_probe_list, out_vals, out_addr, address2id, out_ids, nullptr, nullptr, nullptr,
             nullptr, nullptr, n_slots, nq, max_nprobe, m, k, n_split};
  a.tickets = tickets;
  a.slots_hint = slots_hint;
  return run_packed(a, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int tpq_ivfpq_scan_topk_packed(const uint8_t* packed, const uint8_t* codes,
                                          const float* lut, const uint8_t* is_empty,
                                          const int64_t* cell_start, const int64_t* cell_size,
                                          const int64_t* n_probe_list, float* out_vals,
                                          int64_t* out_addr, const int64_t* address2id,
                                          int64_t* out_ids, int64_t n_slots, int nq, int max_nprobe,
                                          int m, int k, int n_split, void* workspace,
                                          size_t workspace_bytes, tpq_stream_t stream) {
  return tpq_ivfpq_scan_topk_packed_tickets(packed, codes, lut, is_empty, cell_start, cell_size, n_probe_list,
                                            out_vals, out_addr, address2id, out_ids, n_slots, nq, max_nprobe, m,
                                            k, n_split, workspace, workspace_bytes, nullptr, 0, stream);
}

// Which kernels a packed call runs (the order of the tests is run_packed's): TPQ_SCAN_ROUTE_* of torchpq_amd.h.
// `dump_regs`: registers of the finish kernel's exact list on the dump routes.
static int packed_route(const ScanArgs& a, bool residual, int* dump_regs = nullptr) {
  const int m = a.m, k = a.k;
  const int R = list_regs_packed(k);
  if (!has_packed_kernel(m)) return TPQ_SCAN_ROUTE_REF;
  const bool lds_fits =
      scan_lds_bytes_packed(m, R > 16 ? 16 : R, a.max_nprobe, fused_floats_of(a), residual) <= 160 * 1024;
  // k within kBandSlack of 1024 (no room for the candidate band), or a probe table that no longer
  // fits next to the LUT: scan exactly with the reference-layout kernel
  if (R > 16 || !lds_fits) return TPQ_SCAN_ROUTE_REF;
  // (measured against the sorted lists of the three-launch path, C2 shape, 10 000 queries: m = 64, k = 600 / 1000:
  // 6.1 / 6.9 ms against 7.1 / 8.3; m = 120 (1 000 queries), k = 1000: 2.0 against 3.4 -- and k = 600: 1.9 against 1.6;
  // m = 16, 32: within 2 %; k = 300, 500 at m = 64: 4.1 / 4.5 against 3.6 / 4.0)
  const bool pools = !residual && R >= pool_min_list_regs(m) && (packed_waves(m) < 16 || k > 768) && fuse_enabled();
  // large batches: dump_route.  (Where the pools apply as well -- k in (248, 504] of the short codes -- the pools keep the
  // call: dump_route's own k limit for m <= 32, kDumpShortMaxK, says so; variant builds move it for the A/B.)
  {
    const int Rf = dump_finish_regs(k, a.slots_hint);
    if (dump_regs) *dump_regs = Rf;
    const int mode = dump_route(a, residual, Rf);
    if (mode == kDumpSel16 && !pools) return TPQ_SCAN_ROUTE_DUMP_SEL16;
    if (mode == kDumpSel16W8 && !pools) return TPQ_SCAN_ROUTE_DUMP_SEL16_W8;
    if (mode == kDumpF32) return TPQ_SCAN_ROUTE_DUMP_F32;
  }
  if (pools) return TPQ_SCAN_ROUTE_POOLS;
  const int RL = list_regs_scan(k, m, a.max_nprobe, a.slots_hint);
  if (!residual && fuse_enabled() && fuse_fits(m, R) && packed_waves(m) * RL >= R && (a.n_split == 1 || a.tickets))
    return TPQ_SCAN_ROUTE_ONE_LAUNCH;
  return TPQ_SCAN_ROUTE_LISTS;
}

extern "C" int tpq_ivfpq_scan_route(int nq, int k, int n_split, int m, int ds, int max_nprobe, int64_t slots_hint,
                                    int has_lut, int has_packed, int has_tickets, int residual) {
  if (nq <= 0 || k < 1 || k > 1024 || m < 4 || m % 4 != 0 || n_split < 1 || max_nprobe < 1) return -1;
  if (!has_packed) return TPQ_SCAN_ROUTE_REF;
  ScanArgs a{};
  static const float dummy_lut = 0.f;
  static int32_t dummy_tickets = 0;
  a.lut = has_lut ? &dummy_lut : nullptr;   // (only tested against nullptr)
  a.ds = has_lut ? 0 : ds;
  a.nq = nq; a.k = k; a.n_split = n_split; a.m = m; a.max_nprobe = max_nprobe;
  a.slots_hint = slots_hint;
  a.tickets = has_tickets ? &dummy_tickets : nullptr;
  return packed_route(a, residual != 0);
}

static int run_packed(ScanArgs a, const ResidualArgs* ra, void* workspace, size_t workspace_bytes,
                      tpq_stream_t stream) {
  int rc = validate(a);
  if (rc) return rc;
  TPQ_REQUIRE(a.packed != nullptr, "ivfpq_scan_packed: null packed pointer");
  if (a.nq == 0) return TPQ_OK;
  const int nq = a.nq, k = a.k, n_split = a.n_split, m = a.m;
  int32_t* const caller_tickets = a.tickets;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int R = list_regs_packed(k);
  if (!has_packed_kernel(m)) {
    set_error("ivfpq_scan_packed: no scan-layout kernel instantiated for n_subvectors=%d; use "
              "tpq_ivfpq_scan_topk", m);
    return TPQ_ERR_UNSUPPORTED;
  }
  int Rf = 0;
  int route = packed_route(a, ra != nullptr, &Rf);
  if (route == TPQ_SCAN_ROUTE_REF) {
    // k within kBandSlack of 1024 (no room for the candidate band), or a probe table that no longer
    // fits next to the LUT: scan exactly with the reference-layout kernel
    if (ra) {
      a.n_split = 1;
      return run_residual_ref(a, *ra, st);
    }
    const int Rr = list_regs(k);
    if (n_split > 1) {
      rc = need_ws(workspace, workspace_bytes, ws_bytes_for(nq, Rr, n_split), "ivfpq_scan_packed");
      if (rc) return rc;
      fill_ws(a, workspace, Rr, n_split);
    }
    return dispatch_ref(a, Rr, st);
  }
  const int n_lists = n_split * packed_waves(m);
  if (route == TPQ_SCAN_ROUTE_POOLS) {
    // the largest k, plain PQ: pool mode (scan_device.h) -- threshold lists of ceil(k / waves) entries, unsorted pools,
    // one ranking kernel per query; flagged queries (a pool or the ranking buffer overflowed) redone exactly
    // (the ranking kernel takes a query's lists into LDS: fewer workgroups per query when they would not fit)
    if (a.n_split > pool_max_split(m, k)) a.n_split = pool_max_split(m, k);
    const int n_lists_p = a.n_split * packed_waves(m);
    rc = need_ws(workspace, workspace_bytes, pool_ws_bytes(nq, k, m, n_lists_p), "ivfpq_scan_packed");
    if (rc) return rc;
    fill_ws_pool(a, workspace, n_lists_p);
    a.epoch = fresh_epoch();
    a.fuse = 0;
    a.tickets = nullptr;
    a.small_lists = 0;
#ifdef TPQ_SCAN_PROFILE
    a.prof = g_scan_prof;
#endif
    switch (m) {
#define TPQ_CASE_M(M) case M: rc = dispatch_pool_##M(a, pool_list_regs(k, m), st); break;
      TPQ_PACKED_M_LIST(TPQ_CASE_M)
#undef TPQ_CASE_M
      default: rc = TPQ_ERR_UNSUPPORTED; break;
    }
    if (rc) return rc;
    ScanArgs b = a;
    b.n_split = 1;
    b.only_flagged = a.flags;
    return dispatch_ref(b, list_regs(k), st);
  }
  // large batches of plain PQ, k <= 504: dump mode (scan_device.h) -- the scan workgroups stream over the selection table
  // (m = 64: 16-bit, 32 KiB, four workgroups per CU; m = 8, 16, 32: the fp32 table) and end with their lists of fast
  // values, one wave per query finishes
  if (route == TPQ_SCAN_ROUTE_DUMP_SEL16 || route == TPQ_SCAN_ROUTE_DUMP_SEL16_W8 || route == TPQ_SCAN_ROUTE_DUMP_F32) {
    const int mode = route == TPQ_SCAN_ROUTE_DUMP_SEL16 ? kDumpSel16
                                                        : (route == TPQ_SCAN_ROUTE_DUMP_F32 ? kDumpF32 : kDumpSel16W8);
    const int nw = mode == kDumpSel16W8 ? packed_waves(m) : 4;
    const int RLd = list_regs_scan(k, m, a.max_nprobe, a.slots_hint, nw);
    int unsplit = nq, parts = 1;
    if (mode != kDumpSel16W8) dump_tail(nq, nw, RLd, dump_slots(m, mode), &unsplit, &parts);
    // (a caller's workspace sized by an older rule: the tail stays whole)
    if (parts > 1 && (!workspace || workspace_bytes < ws_bytes_for(nq, RLd, parts * nw))) unsplit = nq, parts = 1;
    a.n_split = parts;
    a.unsplit = parts > 1 ? unsplit : 0;  // (0 with one part per query: "all queries split 1 way")
    rc = need_ws(workspace, workspace_bytes, ws_bytes_for(nq, RLd, parts * nw), "ivfpq_scan_packed");
    if (rc) return rc;
    fill_ws(a, workspace, RLd, parts * nw);
    a.epoch = fresh_epoch();
    a.fuse = 0;
    a.tickets = nullptr;
    a.small_lists = 0;
#ifdef TPQ_SCAN_PROFILE
    a.prof = g_scan_prof;
#endif
    switch (m) {
#define TPQ_CASE_M(M) case M: rc = dispatch_dump_##M(a, RLd, Rf, mode, st); break;
      TPQ_PACKED_M_LIST(TPQ_CASE_M)
#undef TPQ_CASE_M
      default: rc = TPQ_ERR_UNSUPPORTED; break;
    }
    if (rc == TPQ_OK) {
      ScanArgs b = a;  // exact redo of the (normally zero) flagged queries
      b.n_split = 1;
      b.only_flagged = a.flags;
      return dispatch_ref(b, list_regs(k), st);
    }
    // TPQ_ERR_UNSUPPORTED = no instantiation for (mode, RLd, Rf) at this m, reported before any launch: the sorted
    // lists below take the call (a change of the list-size heuristics must not fail a search)
    if (rc != TPQ_ERR_UNSUPPORTED) return rc;
    a.n_split = n_split;
    a.unsplit = 0;
    a.tickets = caller_tickets;
    const int RLs = list_regs_scan(k, m, a.max_nprobe, a.slots_hint);
    route = (fuse_enabled() && fuse_fits(m, R) && packed_waves(m) * RLs >= R && (n_split == 1 || a.tickets))
                ? TPQ_SCAN_ROUTE_ONE_LAUNCH : TPQ_SCAN_ROUTE_LISTS;
  }
  // registers of the per-wave lists (<= R)
  const int RL = list_regs_scan(k, m, a.max_nprobe, a.slots_hint);
  a.small_lists = RL < R ? 1 : 0;
  rc = need_ws(workspace, workspace_bytes, ws_bytes_for(nq, RL, n_lists), "ivfpq_scan_packed");
  if (rc) return rc;
  fill_ws(a, workspace, RL, n_lists);
#ifdef TPQ_SCAN_PROFILE
  a.prof = g_scan_prof;
#endif
  // flags: raised == equal to this call's epoch; no zeroing pass (it was a launch of its own)
  a.epoch = fresh_epoch();
  // the scan workgroups finish the query themselves (merge, write, exact redo): one launch instead of three.
  // (k <= 248, plain PQ; a query split over several workgroups needs the caller's tickets)
  a.fuse = route == TPQ_SCAN_ROUTE_ONE_LAUNCH ? 1 : 0;
  if (!a.fuse) a.tickets = nullptr;
  switch (m) {
#define TPQ_CASE_M(M) case M: rc = dispatch_packed_##M(a, ra, RL, R, st); break;
    TPQ_PACKED_M_LIST(TPQ_CASE_M)
#undef TPQ_CASE_M
    default: rc = TPQ_ERR_UNSUPPORTED; break;
  }
  if (rc) return rc;
  if (a.fuse) return TPQ_OK;  // (the fused finish redoes them itself)
  // exact redo of the (normally zero) queries whose candidate band overflowed
  ScanArgs b = a;
  b.n_split = 1;
  b.only_flagged = a.flags;
  if (ra) return run_residual_ref(b, *ra, st);
  return dispatch_ref(b, list_regs(k), st);
}

static int run_residual_ref(ScanArgs a, ResidualArgs ra, hipStream_t st) {
  const int R = list_regs(a.k);
  const bool build_part1 = !ra.full && !ra.part1;
  const size_t lds = scan_lds_bytes_ref(a.m, R, a.max_nprobe, build_part1 ? a.m * a.ds + a.m : 0);
  auto go = [&](auto kernel) -> int {
    int rc2 = set_lds(kernel, lds, "scan_residual_kernel");
    if (rc2) return rc2;
    hipLaunchKernelGGL(kernel, dim3((unsigned)a.nq), dim3(kScanThreads), lds, st, a, ra);
    TPQ_LAUNCH_CHECK("scan_residual_kernel");
    return TPQ_OK;
  };
  switch (R) {
    case 1: return go(scan_residual_kernel<1>);
    case 2: return go(scan_residual_kernel<2>);
    case 4: return go(scan_residual_kernel<4>);
    case 8: return go(scan_residual_kernel<8>);
    default: return go(scan_residual_kernel<16>);
  }
}

extern "C" int tpq_ivfpq_scan_topk_residual(const uint8_t* codes, const float* part1,
                                            const float* part2, const float* full_lut,
                                            const int64_t* cells, const float* base_sims,
                                            const uint8_t* is_empty, const int64_t* cell_start,
                                            const int64_t* cell_size, const int64_t* n_probe_list,
                                            float* out_vals, int64_t* out_addr,
                                            const int64_t* address2id, int64_t* out_ids,
                                            int64_t n_slots, int nq, int max_nprobe, int m, int k,
                                            tpq_stream_t stream) {
  ScanArgs a{codes, nullptr, part1 ? part1 : full_lut, nullptr, nullptr, 0, 0, is_empty, cell_start,
             cell_size, n_probe_list, out_vals, out_addr, address2id, out_ids, nullptr, nullptr,
             nullptr, nullptr, nullptr, n_slots, nq, max_nprobe, m, k, 1};
  int rc = validate(a);
  if (rc) return rc;
  TPQ_REQUIRE(base_sims != nullptr, "ivfpq_scan_residual: base_sims is required");
  TPQ_REQUIRE(full_lut != nullptr || (part1 && part2 && cells),
              "ivfpq_scan_residual: need either full_lut or (part1, part2, cells)");
  if (nq == 0) return TPQ_OK;
  ResidualArgs ra{part1, part2, full_lut, cells, base_sims, nullptr, nullptr};
  return run_residual_ref(a, ra, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int tpq_ivfpq_scan_topk_residual_packed(
    const uint8_t* packed, const uint8_t* codes, const float* part1, const float* query,
    const float* codebook, int ds, const float* part2, const float* slot_term,
    const float* cell_bound, const int64_t* cells, const float* base_sims, const uint8_t* is_empty,
    const int64_t* cell_start, const int64_t* cell_size, const int64_t* n_probe_list,
    float* out_vals, int64_t* out_addr, const int64_t* address2id, int64_t* out_ids,
    int64_t n_slots, int nq, int max_nprobe, int m, int k, int n_split, void* workspace,
    size_t workspace_bytes, tpq_stream_t stream) {
  TPQ_REQUIRE(part1 || (query && codebook && ds >= 1 && ds <= 1024),
              "ivfpq_scan_residual_packed: need part1 or (query, codebook, ds)");
  TPQ_REQUIRE(part2 && slot_term && cell_bound && cells && base_sims,
              "ivfpq_scan_residual_packed: null pointer argument");
  // euclid = 2: the in-workgroup LUT is part1 = 2 q_j.r_jc
  ScanArgs a{codes, packed, part1, query, codebook, ds, 2, is_empty, cell_start, cell_size,
             n_probe_list, out_vals, out_addr, address2id, out_ids, nullptr, nullptr, nullptr,
             nullptr, nullptr, n_slots, nq, max_nprobe, m, k, n_split};
  ResidualArgs ra{part1, part2, nullptr, cells, base_sims, slot_term, cell_bound};
  return run_packed(a, &ra, workspace, workspace_bytes, stream);
}

// ---- per-slot / per-cell constants of the packed residual scan ----------------------------------
namespace tpq {
// slot_term[s] = sum_j part2[cell(s)][j][code_j(s)] (ascending j, fp32) for the slots inside
// [cell_start, cell_start + cell_size); cell_bound[c] = sum_j max_c' |part2[c][j][c']|
__global__ __launch_bounds__(256) void residual_slot_term_kernel(
    const uint32_t* __restrict__ codes32, const float* __restrict__ part2,
    const int64_t* __restrict__ cell_start, const int64_t* __restrict__ cell_size,
    float* __restrict__ slot_term, int64_t n_slots, int m) {
  const int cell = blockIdx.x;
  const float* __restrict__ p2 = part2 + (int64_t)cell * m * 256;
  const int64_t st = cell_start[cell], sz = cell_size[cell];
  const int G = m >> 2;
  for (int64_t o = (int64_t)blockIdx.y * 256 + threadIdx.x; o < sz; o += (int64_t)gridDim.y * 256) {
    const int64_t s = st + o;
    float v = 0.f;
    for (int g = 0; g < G; ++g) {
      const uint32_t w = codes32[(int64_t)g * n_slots + s];
      const float* row = p2 + g * 1024;
      v += row[w & 255u];
      v += row[256 + ((w >> 8) & 255u)];
      v += row[512 + ((w >> 16) & 255u)];
      v += row[768 + (w >> 24)];
    }
    slot_term[s] = v;
  }
}

__global__ __launch_bounds__(64) void residual_cell_bound_kernel(const float* __restrict__ part2,
                                                                 float* __restrict__ cell_bound,
                                                                 int m) {
  const int cell = blockIdx.x, lane = threadIdx.x;
  const float4* __restrict__ p2 = reinterpret_cast<const float4*>(part2 + (int64_t)cell * m * 256);
  float sum = 0.f;
  for (int j = 0; j < m; ++j) {
    const float4 x = p2[j * 64 + lane];
    float mx = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    sum += mx;
  }
  if (lane == 0) cell_bound[cell] = sum;
}
}  // namespace tpq

extern "C" int tpq_ivfpq_residual_slot_terms(const uint8_t* codes, const float* part2,
                                             const int64_t* cell_start, const int64_t* cell_size,
                                             float* slot_term, float* cell_bound, int64_t n_slots,
                                             int n_cells, int m, tpq_stream_t stream) {
  TPQ_REQUIRE(codes && part2 && cell_start && cell_size && slot_term && cell_bound,
              "ivfpq_residual_slot_terms: null pointer argument");
  TPQ_REQUIRE(m >= 4 && m % 4 == 0 && n_cells >= 1 && n_slots >= 0 && n_slots < 0x7fffffffLL,
              "ivfpq_residual_slot_terms: bad shape (m=%d, n_cells=%d, n_slots=%lld)", m, n_cells,
              (long long)n_slots);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = check_hip(hipMemsetAsync(slot_term, 0, (size_t)n_slots * 4, st), "slot_terms memset");
  if (rc) return rc;
  hipLaunchKernelGGL(residual_cell_bound_kernel, dim3(n_cells), dim3(64), 0, st, part2, cell_bound, m);
  TPQ_LAUNCH_CHECK("residual_cell_bound_kernel");
  if (n_slots == 0) return TPQ_OK;
  const int per_cell = (int)((n_slots / n_cells + 255) / 256);
  const int by = per_cell < 1 ? 1 : (per_cell > 64 ? 64 : per_cell);
  hipLaunchKernelGGL(residual_slot_term_kernel, dim3(n_cells, by), dim3(256), 0, st,
                     reinterpret_cast<const uint32_t*>(codes), part2, cell_start, cell_size,
                     slot_term, n_slots, m);
  TPQ_LAUNCH_CHECK("residual_slot_term_kernel");
  return TPQ_OK;
}

extern "C" int tpq_residual_part1(const float* query, const float* codebook, float* part1, int m,
                                  int ds, int nq, tpq_stream_t stream) {
  TPQ_REQUIRE(query && codebook && part1, "residual_part1: null pointer");
  TPQ_REQUIRE(m >= 1 && m <= 65535 && ds >= 1 && nq >= 0, "residual_part1: bad shape");
  if (nq == 0) return TPQ_OK;
  hipLaunchKernelGGL(residual_part1_kernel, dim3(nq, m), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), query, codebook, part1, m, ds, nq);
  TPQ_LAUNCH_CHECK("residual_part1_kernel");
  return TPQ_OK;
}
