// Scan-layout list scan: one translation unit per sub-quantizer count, compiled with
// -DTPQ_PACKED_M=<M> (build.sh) so the 5 x 2 x 5 kernel instantiations build in parallel.
// Device code: scan_device.h; C ABI: scan.hip.
#include "scan_device.h"

#ifndef TPQ_PACKED_M
#error "compile with -DTPQ_PACKED_M=<n_subvectors>"
#endif

namespace tpq {

template <int RL, int R, int M, bool RES>
static int launch_packed(ScanArgs a, ResidualArgs ra, hipStream_t st) {
  size_t lds = scan_lds_bytes_packed(M, RL, a.max_nprobe, fused_floats_of(a), RES);
#ifdef TPQ_EXTRA_LDS  // experiment: force fewer workgroups per CU
  lds += TPQ_EXTRA_LDS;
#endif
  if constexpr (!RES && R <= kFuseMaxR) {
    if (a.fuse) {  // fused finish: one launch (scan.hip decides)
      int rc = set_lds(scan_packed_kernel<RL, M, false, R>, lds, "scan_packed_kernel (fused finish)");
      if (rc) return rc;
      const float delta_rel = 1.05f * 2.0f * 5.9604645e-8f * (float)(M - 1);
      hipLaunchKernelGGL((scan_packed_kernel<RL, M, false, R>), dim3((unsigned)a.nq * a.n_split),
                         dim3(packed_waves(M) * 64), lds, st, a, ra, delta_rel);
      TPQ_LAUNCH_CHECK("scan_packed_kernel (fused finish)");
      return TPQ_OK;
    }
  }
  int rc = set_lds(scan_packed_kernel<RL, M, RES>, lds, "scan_packed_kernel");
  if (rc) return rc;
  // delta = 1.05 * 2 (M-1) u * sum_j max|LUT_j|,  u = 2^-24   (residual: M+1 roundings, see kernel)
  const float delta_rel = 1.05f * 2.0f * 5.9604645e-8f * (float)(RES ? M + 1 : M - 1);
  hipLaunchKernelGGL((scan_packed_kernel<RL, M, RES>), dim3((unsigned)a.nq * a.n_split),
                     dim3(packed_waves(M) * 64), lds, st, a, ra, delta_rel);
  TPQ_LAUNCH_CHECK("scan_packed_kernel");
  const int n_lists = a.n_split * packed_waves(M);
  // merge waves per query: the largest power of two <= min(8, n_lists / 2) that divides n_lists
  // (n_lists = n_split x 4 / 8 / 16 waves: 2, 4 or 8)
  int W = 8;
  while (W > 1 && (W > n_lists / 2 || n_lists % W != 0)) W >>= 1;
  const size_t merge_lds = (size_t)W * R * 64 * 8;
  rc = set_lds(scan_merge_refine_kernel<RL, R, M, RES>, merge_lds, "scan_merge_refine_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL((scan_merge_refine_kernel<RL, R, M, RES>), dim3(a.nq), dim3(W * 64), merge_lds, st, a);
  TPQ_LAUNCH_CHECK("scan_merge_refine_kernel");
  return TPQ_OK;
}

// pool mode (k > 248): scan with threshold lists of RL registers + pools (RM = -1: 1 024 entries per wave, -2: 2 048),
// then the ranking kernel
template <int RL, int M, int RM>
static int launch_pool(ScanArgs a, hipStream_t st) {
  const size_t lds = scan_lds_bytes_packed(M, RL, a.max_nprobe, fused_floats_of(a), false);
  int rc = set_lds(scan_packed_kernel<RL, M, false, RM>, lds, "scan_packed_kernel (pool mode)");
  if (rc) return rc;
  const float delta_rel = 1.05f * 2.0f * 5.9604645e-8f * (float)(M - 1);
  hipLaunchKernelGGL((scan_packed_kernel<RL, M, false, RM>), dim3((unsigned)a.nq * a.n_split),
                     dim3(packed_waves(M) * 64), lds, st, a, ResidualArgs{}, delta_rel);
  TPQ_LAUNCH_CHECK("scan_packed_kernel (pool mode)");
  constexpr int NW = packed_waves(M), LEN = 64 * (NW == 4 ? 8 : 4);
  const size_t mlds = (size_t)a.n_split * NW * LEN * 8 + (size_t)((a.k + 63) / 64 * 64) * 8;
  rc = set_lds(scan_pool_merge_kernel<NW>, mlds, "scan_pool_merge_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL((scan_pool_merge_kernel<NW>), dim3(a.nq), dim3(kPoolMergeThreads), mlds, st, a);
  TPQ_LAUNCH_CHECK("scan_pool_merge_kernel");
  return TPQ_OK;
}

// dump modes (scan_device.h): the scan ends with the waves' lists of fast values; scan_finish_exact_kernel, one wave per
// query, evaluates the band's survivors exactly and writes the result.  m = 64: the 16-bit table (kDumpSel16 / W8);
// m = 8, 16, 32 (round 6): the fp32 table the four-wave workgroups of the short codes stream over anyway (kDumpF32) --
// what they gain is the early end of the scan workgroup.
constexpr bool has_sel16(int M) { return M == 64; }
constexpr bool has_dump_f32(int M) { return M == 8 || M == 16 || M == 32; }
constexpr bool has_dump(int M) { return has_sel16(M) || has_dump_f32(M); }
// the finish kernel: survivors in RM = 2, 4 or 8 registers (k <= 56 rides on 2); the exact entries from the codebook
// in LDS (fused calls; sub-vector length 1 / 2 compiled in at m = 64, read from the arguments otherwise) or gathered from
// the caller's table (m <= 32); 4, 8 or 16 chunks of 64 keys per query
template <int RM, int M>
static int launch_finish(const ScanArgs& a, int nw_scan, int RL, hipStream_t st) {
  if constexpr (!has_dump(M)) {
    return TPQ_ERR_UNSUPPORTED;
  } else {
    const bool from_lut = a.lut != nullptr;
    const size_t flds = finish_lds_bytes(M, a.ds, RM, from_lut);
    // one persistent workgroup per CU (its LDS holds the codebook)
    int dev = 0, n_cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0)
      n_cus = 256;
    constexpr int FW = finish_waves(RM);
    const int want = (a.nq + FW - 1) / FW;
    // (the table-gathering form holds no codebook: workgroups are cheap, two per CU hide the gathers' latency)
    const int cap = from_lut ? 2 * n_cus : n_cus;
    const dim3 grid((unsigned)(want < cap ? want : cap)), block(FW * 64);
    auto go = [&](auto kernel) -> int {
      int rc = set_lds(kernel, flds, "scan_finish_exact_kernel");
      if (rc) return rc;
      hipLaunchKernelGGL(kernel, grid, block, flds, st, a, nw_scan, RL);
      TPQ_LAUNCH_CHECK("scan_finish_exact_kernel");
      return TPQ_OK;
    };
    const int T = a.n_split * nw_scan * RL;  // (16: the split tail of a batch, ScanArgs::unsplit)
    if constexpr (has_sel16(M)) {
      if (a.ds == 1)
        return T <= 4 ? go(scan_finish_exact_kernel<RM, M, 1, 4>)
                      : (T <= 8 ? go(scan_finish_exact_kernel<RM, M, 1, 8>) : go(scan_finish_exact_kernel<RM, M, 1, 16>));
      return T <= 4 ? go(scan_finish_exact_kernel<RM, M, 2, 4>)
                    : (T <= 8 ? go(scan_finish_exact_kernel<RM, M, 2, 8>) : go(scan_finish_exact_kernel<RM, M, 2, 16>));
    } else {
      if (from_lut)
        return T <= 4 ? go(scan_finish_exact_kernel<RM, M, 0, 4, true>)
                      : (T <= 8 ? go(scan_finish_exact_kernel<RM, M, 0, 8, true>)
                                : go(scan_finish_exact_kernel<RM, M, 0, 16, true>));
      // (the sub-vector length of a 128-dimensional index compiled in: SIFT's m = 32 -> ds = 4; others read it)
      constexpr int DSF = 128 / M;
      if (a.ds == DSF)
        return T <= 4 ? go(scan_finish_exact_kernel<RM, M, DSF, 4>)
                      : (T <= 8 ? go(scan_finish_exact_kernel<RM, M, DSF, 8>)
                                : go(scan_finish_exact_kernel<RM, M, DSF, 16>));
      return T <= 4 ? go(scan_finish_exact_kernel<RM, M, 0, 4>)
                    : (T <= 8 ? go(scan_finish_exact_kernel<RM, M, 0, 8>) : go(scan_finish_exact_kernel<RM, M, 0, 16>));
    }
  }
}

template <int RL, int R, int M, int MODE>
static int launch_dump(ScanArgs a, hipStream_t st) {
  if constexpr ((is_sel16(MODE) && !has_sel16(M)) || (MODE == kDumpF32 && !has_dump_f32(M)) ||
                (MODE == kDumpSel16W8 && RL > 2)) {
    return TPQ_ERR_UNSUPPORTED;  // (no message: scan.hip falls back to the sorted-list paths)
  } else {
    const size_t lds = scan_lds_bytes_dump(M, is_sel16(MODE), scan_waves(M, MODE), a.max_nprobe, fused_floats_of(a));
    int rc = set_lds(scan_packed_kernel<RL, M, false, MODE>, lds, "scan_packed_kernel (dump mode)");
    if (rc) return rc;
    const float delta_rel = 1.05f * 2.0f * 5.9604645e-8f * (float)(M - 1);
    constexpr int NW = scan_waves(M, MODE);
    const unsigned blocks = (unsigned)a.unsplit + (unsigned)(a.nq - a.unsplit) * (unsigned)a.n_split;
    hipLaunchKernelGGL((scan_packed_kernel<RL, M, false, MODE>), dim3(blocks), dim3(NW * 64), lds, st, a,
                       ResidualArgs{}, delta_rel);
    TPQ_LAUNCH_CHECK("scan_packed_kernel (dump mode)");
    return launch_finish<(R < 2 ? 2 : R), M>(a, NW, RL, st);
  }
}
// TPQ_ERR_UNSUPPORTED without a launch = "no instantiation for these list registers": the caller (scan.hip) falls
// through to the sorted-list paths instead of failing the search (the pairs below cover what
// list_regs_scan / dump_finish_regs produce today; a change of either heuristic must not turn into an error)
template <int M, int MODE>
static int dispatch_dump_mode(const ScanArgs& a, int RL, int R, hipStream_t st) {
#define TPQ_PAIR(A, B) if (RL == A && R == B) return launch_dump<A, B, M, MODE>(a, st);
  if constexpr (MODE == kDumpSel16W8) {
    TPQ_PAIR(1, 8) TPQ_PAIR(2, 8) TPQ_PAIR(2, 16)
  } else if constexpr (MODE == kDumpF32) {
    TPQ_PAIR(1, 1) TPQ_PAIR(1, 2) TPQ_PAIR(2, 2) TPQ_PAIR(1, 4) TPQ_PAIR(2, 4) TPQ_PAIR(4, 4) TPQ_PAIR(2, 8) TPQ_PAIR(4, 8)
    TPQ_PAIR(4, 16)
  } else {
    TPQ_PAIR(1, 1) TPQ_PAIR(1, 2) TPQ_PAIR(2, 2) TPQ_PAIR(1, 4) TPQ_PAIR(2, 4) TPQ_PAIR(2, 8) TPQ_PAIR(4, 8)
  }
#undef TPQ_PAIR
  return TPQ_ERR_UNSUPPORTED;
}

template <int M, bool RES>
static int dispatch_r(const ScanArgs& a, const ResidualArgs& ra, int RL, int R, hipStream_t st) {
  if (RL == R) {
    switch (R) {
      case 1: return launch_packed<1, 1, M, RES>(a, ra, st);
      case 2: return launch_packed<2, 2, M, RES>(a, ra, st);
      case 4: return launch_packed<4, 4, M, RES>(a, ra, st);
      case 8: return launch_packed<8, 8, M, RES>(a, ra, st);
      default: return launch_packed<16, 16, M, RES>(a, ra, st);
    }
  }
  // short per-wave lists (list_regs_scan): every (RL < R) pair
#define TPQ_PAIR(A, B) if (RL == A && R == B) return launch_packed<A, B, M, RES>(a, ra, st);
  TPQ_PAIR(1, 2) TPQ_PAIR(1, 4) TPQ_PAIR(1, 8) TPQ_PAIR(1, 16) TPQ_PAIR(2, 4) TPQ_PAIR(2, 8)
  TPQ_PAIR(2, 16) TPQ_PAIR(4, 8) TPQ_PAIR(4, 16) TPQ_PAIR(8, 16)
#undef TPQ_PAIR
  set_error("scan_packed: no instantiation for list registers (%d, %d)", RL, R);
  return TPQ_ERR_UNSUPPORTED;
}

#define TPQ_CAT2(a, b) a##b
#define TPQ_CAT(a, b) TPQ_CAT2(a, b)

int TPQ_CAT(dispatch_pool_, TPQ_PACKED_M)(const ScanArgs& a, int RL, hipStream_t st) {
  const bool big = a.pool_cap > 1024;
  switch (RL) {
    // (-3: the large pool without the counting rounds -- see kPoolRoundsFromK)
#define TPQ_POOL_CASE(RL)                                                                              \
  case RL:                                                                                             \
    return !big ? launch_pool<RL, TPQ_PACKED_M, -1>(a, st)                                             \
                : (a.k <= kPoolRoundsFromK ? launch_pool<RL, TPQ_PACKED_M, -3>(a, st)                  \
                                           : launch_pool<RL, TPQ_PACKED_M, -2>(a, st));
    TPQ_POOL_CASE(1)
    TPQ_POOL_CASE(2)
    TPQ_POOL_CASE(4)
#undef TPQ_POOL_CASE
    default: break;
  }
  set_error("scan_packed (pool mode): no instantiation for %d list registers", RL);
  return TPQ_ERR_UNSUPPORTED;
}

// returns TPQ_ERR_UNSUPPORTED -- before anything was launched -- when (mode, RL, R) has no instantiation at this m
int TPQ_CAT(dispatch_dump_, TPQ_PACKED_M)(const ScanArgs& a, int RL, int R, int mode, hipStream_t st) {
  if (mode == kDumpSel16W8) return dispatch_dump_mode<TPQ_PACKED_M, kDumpSel16W8>(a, RL, R, st);
  if (mode == kDumpF32) return dispatch_dump_mode<TPQ_PACKED_M, kDumpF32>(a, RL, R, st);
  return dispatch_dump_mode<TPQ_PACKED_M, kDumpSel16>(a, RL, R, st);
}

// workgroups of the dump-mode scan kernel (RL = 1, a 64-probe table, fused LUT at ds = 2) one CU holds; 0 = not built
template <int M, int MODE>
static int dump_occupancy_of() {
  if constexpr ((is_sel16(MODE) && !has_sel16(M)) || (MODE == kDumpF32 && !has_dump_f32(M))) {
    return 0;
  } else {
    constexpr int NW = scan_waves(M, MODE);
    const size_t lds = scan_lds_bytes_dump(M, is_sel16(MODE), NW, 64, M * 2 + M);
    int n = 0;
    if (set_lds(scan_packed_kernel<1, M, false, MODE>, lds, "scan_packed_kernel (dump mode)") != TPQ_OK) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_packed_kernel<1, M, false, MODE>, NW * 64, lds) != hipSuccess)
      return 0;
    return n;
  }
}
int TPQ_CAT(dump_occupancy_, TPQ_PACKED_M)(int mode) {
  if (mode == kDumpSel16W8) return dump_occupancy_of<TPQ_PACKED_M, kDumpSel16W8>();
  if (mode == kDumpF32) return dump_occupancy_of<TPQ_PACKED_M, kDumpF32>();
  return dump_occupancy_of<TPQ_PACKED_M, kDumpSel16>();
}

int TPQ_CAT(dispatch_packed_, TPQ_PACKED_M)(const ScanArgs& a, const ResidualArgs* ra, int RL, int R,
                                            hipStream_t st) {
  if (ra) return dispatch_r<TPQ_PACKED_M, true>(a, *ra, RL, R, st);
  return dispatch_r<TPQ_PACKED_M, false>(a, ResidualArgs{}, RL, R, st);
}

}  // namespace tpq
