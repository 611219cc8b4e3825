// Device code of the IVF list scan (shared by scan.hip and the per-M scan_packed.hip units).
#pragma once
#include "common.h"
#include "scan_layout.h"
#include "wave_topk.h"

namespace tpq {

#ifndef TPQ_LUT_U
#define TPQ_LUT_U 4  // fused LUT build, ds <= 2: entries (float4 groups) per thread whose codebook loads are issued together
#endif
// pool mode: the counting rounds that tighten the cut before the exact evaluation pay beyond this k (same box, m = 64, ms
// per 10 000 queries, 0 / 1 / 3 rounds COMPILED IN: k = 600: 4.79 / 5.14 / 5.27, k = 800: 5.19 / 5.50 / 5.68, k = 1000:
// 6.66 / 6.70 / 6.08; three rounds compiled in and none executed: 5.48 at k = 600 -- hence a kernel of its own, RM = -3)
constexpr int kPoolRoundsFromK = 900;
// pool mode from list_regs_packed(k) = 16 on, i.e. k > 504 (eight waves, from k > 248 with pools of 1 024 and no rounds,
// same box, ms per 10 000 queries, lists -> pool: m = 64, k = 300 / 400 / 500: 3.57 / 3.79 / 3.96 -> 3.76 / 3.95 / 4.09)
constexpr int kPoolMinListRegs = 16;
// ... and, with four waves per workgroup (m <= 32), from list_regs_packed(k) = 8 on (k > 248, where the fused finish of the
// sorted lists ends): lists -> pools of 2 048 without rounds, same box, ms per 10 000 queries: m = 32, k = 300 / 400 / 500:
// 3.49 / 3.72 / 3.96 -> 2.67 / 2.83 / 2.97; m = 16, k = 300 / 500: 2.27 / 2.55 -> 1.65 / 1.82; m = 8, k = 400: 2.32 -> 1.45;
// IVF4096 cells, m = 32, k = 400: 3.33 -> 1.99.  (k <= 248 stays with the lists: m = 32, k = 248: 2.04 against 2.40.)
static int pool_min_list_regs(int m) { return m <= 32 ? 8 : kPoolMinListRegs; }
constexpr int kScanWaves = 8;
constexpr int kScanThreads = kScanWaves * 64;

// ws_delta[q] after a call: the value scan_ref_kernel / scan_residual_kernel leave when they redo a FLAGGED query (a
// selection band is >= 0 and the one-launch finisher writes 1.f / 0.f: -1 is neither) -- IVFPQTopkHip.last_redone
constexpr float kRedoneMark = -1.f;

struct ScanArgs {
  const uint8_t* codes;    // reference layout [m/4][n_slots][4]
  const uint8_t* packed;   // scan layout (packed kernel only)
  const float* lut;        // [m][nq][256]; nullptr = build the LUT in the workgroup ("fused")
  const float* query;      // fused: [m*ds][nq]
  const float* codebook;   // fused: [m][ds][256]
  int ds, euclid;          // fused: sub-vector length, 1 = 2ab-a^2-b^2 / 0 = dot / 2 = 2ab
  const uint8_t* is_empty; // nullable
  const int64_t* cell_start;
  const int64_t* cell_size;
  const int64_t* n_probe_list;
  float* out_vals;
  int64_t* out_addr;
  const int64_t* address2id;
  int64_t* out_ids;
  float* ws_vals;  // [nq][n_split][64R]
  int* ws_idx;
  int* flags;             // [nq] packed path: == epoch: candidate band overflowed, redo exactly.  Never zeroed:
                          // "raised" is equality with this call's epoch (the workspace arrives as garbage; a
                          // word that happens to equal the epoch costs one needless exact redo, never a wrong result)
  float* ws_delta;        // [nq] packed path: fast-vs-exact error bound of the query
  const int* only_flagged;  // reference kernel: when set, only queries with a non-zero flag run
  int64_t n_slots;
  int nq, max_nprobe, m, k, n_split;
  unsigned long long* prof;  // -DTPQ_SCAN_PROFILE builds: [nq][16] phase timestamps (10 ns ticks)
  int small_lists;           // packed path, large k: per-wave lists hold fewer than k + 8 entries
  int epoch;                 // value that marks a raised flag in this call (non-zero)
  int* tickets;              // fused finish, n_split > 1: the CALLER's [nq] int32, zero on entry, zero on exit
  int fuse;                  // fused finish (scan_packed_kernel RM > 0): the scan workgroups write the result
  int64_t slots_hint;        // host only: expected slots scanned per query (0 = unknown), sizes the per-wave lists
  // pool mode (k > 248, scan_packed_kernel RM < 0): per (query, split, wave) an append-only pool of pool_cap
  // admitted candidates (keys: value image, ~address), later overwritten in place by the exact candidates
  unsigned* pool_hi;
  unsigned* pool_lo;
  int* pool_cnt;             // [nq][n_lists] exact candidates the list holds after the scan kernel
  int pool_cap;
  // dump mode (scan_packed_kernel RM <= kDumpF32): [nq][n_lists] 1 = the wave's list may have evicted a candidate
  int* list_evict;
  // dump mode, "tail split": queries [0, unsplit) run as ONE workgroup each, queries [unsplit, nq) as n_split
  // workgroups each (0 = every query is split n_split ways, the meaning of n_split everywhere else).  A batch that is
  // not a multiple of the chip's workgroup slots ends with a round of few workgroups, each as long as a whole query
  // (1 250 queries on 1 024 slots: two rounds for 1.22 rounds of work); the queries of that last round are dealt
  // as short workgroups instead -- they start last (workgroups are dispatched in index order) and fill the slots
  // the long ones leave.  The lists keep the stride of n_split parts for every query.
  int unsplit;
};

// scan_packed_kernel modes beyond the fused finish (RM > 0) and the pools (RM = -1, -2, -3): "dump" -- the scan
// workgroup ends with its waves' lists of FAST values; scan_finish_exact_kernel (one wave per query, full occupancy)
// merges them, evaluates the band's survivors exactly from global memory and writes the result.
constexpr int kDumpF32 = -8;     // fp32 table in LDS (m KiB), the permuted-order fp32 sum as the selection key
constexpr int kDumpSel16 = -16;  // 16-bit fixed-point table (m / 2 KiB), an exact integer sum as the selection key
constexpr int kDumpSel16W8 = -17;  // the same with the eight waves of the other paths (k in (248, 504]: lists of <= 2 registers)
constexpr bool is_sel16(int RM) { return RM == kDumpSel16 || RM == kDumpSel16W8; }
constexpr int kDumpMinQueries = 1024;  // batches that fill the chip's 4 x 256 workgroup slots at least once
constexpr int kDumpShortMaxK = 248;    // m = 8, 16, 32 (kDumpF32): the pools take the larger k
constexpr int kDumpLutMinSlots = 24576;  // ... with a caller's table: from this many expected slots per query on

#ifdef TPQ_SCAN_PROFILE
#define TPQ_PROF(a, q, i)                                                        \
  do {                                                                           \
    if ((a).prof && threadIdx.x == 0) (a).prof[(int64_t)(q) * 16 + (i)] = wall_clock64(); \
  } while (0)
#else
#define TPQ_PROF(a, q, i) ((void)0)
#endif

// Volatile accesses to LDS words other waves update (the shared admission threshold, the waves' quantiles) go through an
// LDS-ADDRESS-SPACE pointer.  A `volatile T*` cast of a generic pointer compiles to FLAT loads, and FLAT counts on vmcnt:
// the `s_waitcnt vmcnt(0)` hipcc put behind the per-tile threshold poll made every wave wait, once per tile, until the NEXT
// tile's code loads -- the software pipeline's prefetch, issued a few hundred cycles earlier -- had landed (round 6, read
// off the ISA of the tile loop: `flat_load_dword ... sc0 sc1` + `s_waitcnt vmcnt(0)`).  ds_read_b32 counts on lgkmcnt only.
__device__ __forceinline__ unsigned lds_poll_u32(const unsigned* p) {
  typedef const volatile __attribute__((address_space(3))) unsigned* lds_ptr;
  return *(lds_ptr)p;
}
__device__ __forceinline__ float lds_poll_f32(const float* p) {
  typedef const volatile __attribute__((address_space(3))) float* lds_ptr;
  return *(lds_ptr)p;
}
__device__ __forceinline__ void lds_post_f32(float* p, float v) {
  typedef volatile __attribute__((address_space(3))) float* lds_ptr;
  *(lds_ptr)p = v;
}

// ---- shared pieces -----------------------------------------------------------------------

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt [3:0] and [15:14], expcnt [6:4] and lgkmcnt [11:8] left at their maxima)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "six bits");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

struct ProbeTable {  // lives in LDS
  int* start;        // [max_nprobe]
  int* size;         // [max_nprobe]
  int* tile_begin;   // [max_nprobe + 1] exclusive prefix of ceil(size/64)
};

// wave 0 fills the probe table; cells whose start equals the previous probe's start are
// skipped (ivfpq_topk.cu:864-866)
struct ProbeRegs {  // the first 64 probes' extents, one per lane (fetch_probes: the loads are issued early)
  int st, sz;
};
__device__ __forceinline__ ProbeRegs fetch_probes(const ScanArgs& a, int q, int n_probe, int base) {
  const int p = base + lane_id();
  ProbeRegs r{0, 0};
  if (p < n_probe) {
    r.st = (int)a.cell_start[(int64_t)q * a.max_nprobe + p];
    r.sz = (int)a.cell_size[(int64_t)q * a.max_nprobe + p];
    if (p > 0 && a.cell_start[(int64_t)q * a.max_nprobe + p - 1] == (int64_t)r.st) r.sz = 0;
    if (r.sz < 0) r.sz = 0;
  }
  return r;
}
__device__ __forceinline__ void build_probe_table(const ScanArgs& a, int q, int n_probe,
                                                  ProbeTable t, int tile_shift = 6,
                                                  const ProbeRegs* first = nullptr) {
  const int lane = lane_id();
  int running = 0;
  for (int base = 0; base < n_probe; base += 64) {
    const int p = base + lane;
    const ProbeRegs r = (base == 0 && first) ? *first : fetch_probes(a, q, n_probe, base);
    const int st = r.st, sz = r.sz;
    int tiles = (sz + (1 << tile_shift) - 1) >> tile_shift;
    int incl = tiles;  // inclusive wave scan
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (p < n_probe) {
      t.start[p] = st;
      t.size[p] = sz;
      t.tile_begin[p] = running + incl - tiles;
    }
    running += readlane_i(incl, 63);
  }
  if (lane == 0) t.tile_begin[n_probe] = running;
}

// lists travel as keys: `lv` holds the high words (value images), `li` the low words (~index)
template <int R>
__device__ __forceinline__ void store_list(const WaveTopK<R>& top, float* lv, int* li) {
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    reinterpret_cast<unsigned*>(lv)[r * 64 + lane] = top.k[r].hi;
    reinterpret_cast<unsigned*>(li)[r * 64 + lane] = top.k[r].lo;
  }
}

template <int R>
__device__ __forceinline__ void merge_list(WaveTopK<R>& top, const float* lv, const int* li) {
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < R; ++r)
    top.insert_sorted(Key{reinterpret_cast<const unsigned*>(lv)[r * 64 + lane],
                          reinterpret_cast<const unsigned*>(li)[r * 64 + lane]});
}

template <int R>
__device__ __forceinline__ void write_final(const ScanArgs& a, int q, const WaveTopK<R>& top) {
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * 64 + lane;
    if (e < a.k) {
      const int idx = key_index(top.k[r]);
      const bool pad = (idx == kPadIdx);
      const int64_t adr = pad ? -1 : (int64_t)idx;
      a.out_vals[(int64_t)q * a.k + e] = pad ? -INFINITY : key_value(top.k[r]);
      a.out_addr[(int64_t)q * a.k + e] = adr;
      if (a.out_ids) a.out_ids[(int64_t)q * a.k + e] = pad ? -1 : a.address2id[adr];
    }
  }
}

// Cross-wave tree merge through LDS (`lv`/`li` may alias the dead LUT), then output.
template <int R>
__device__ __forceinline__ void finish_query(const ScanArgs& a, int q, int part,
                                             WaveTopK<R>& top, float* lv, int* li) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // every wave is done with the LUT
  for (int stride = 1; stride < kScanWaves; stride <<= 1) {
    if ((wave & (2 * stride - 1)) == stride) store_list<R>(top, lv + wave * R * 64, li + wave * R * 64);
    __syncthreads();
    if ((wave & (2 * stride - 1)) == 0)
      merge_list<R>(top, lv + (wave + stride) * R * 64, li + (wave + stride) * R * 64);
    __syncthreads();
  }
  if (wave == 0) {
    if (a.n_split == 1) {
      write_final<R>(a, q, top);
    } else {
      const int64_t o = ((int64_t)q * a.n_split + part) * (R * 64);
      store_list<R>(top, a.ws_vals + o, a.ws_idx + o);
    }
  }
}

// ---- LUT built inside the workgroup ("fused") ------------------------------------------------
// Instead of reading a materialised [m][nq][256] table (a-3 writes 655 MB and the scan reads it
// back at C2), the workgroup computes its query's LUT from the query and the PQ codebook, which
// stays L2-resident (m*ds KiB).  The arithmetic is adc_lut_kernel's, operation for operation --
// dot, |q|^2 and |c|^2 as ascending-dimension fma chains, then 2*dot, -|q|^2, -|c|^2 -- so the
// entries are bit-identical to tpq_adc_lut's.
__device__ __forceinline__ void stage_query(const ScanArgs& a, int q, float* xq, int n_threads) {
  const int d = a.m * a.ds;
  for (int i = threadIdx.x; i < d; i += n_threads) xq[i] = a.query[(int64_t)i * a.nq + q];
  __syncthreads();
}

__device__ __forceinline__ float4 fused_lut4(const ScanArgs& a, int j, int c4, const float* xq) {
  const float4* __restrict__ cb = reinterpret_cast<const float4*>(a.codebook) + (int64_t)j * a.ds * 64 + c4;
  float4 dot = make_float4(0.f, 0.f, 0.f, 0.f), c2 = dot;
  float q2 = 0.f;  // |q_j|^2, the same ascending-dimension chain in every thread that needs it
  for (int e = 0; e < a.ds; ++e) {
    const float4 y = cb[e * 64];
    const float x = xq[j * a.ds + e];
    q2 = fmaf(x, x, q2);
    dot.x = fmaf(x, y.x, dot.x); dot.y = fmaf(x, y.y, dot.y);
    dot.z = fmaf(x, y.z, dot.z); dot.w = fmaf(x, y.w, dot.w);
    c2.x = fmaf(y.x, y.x, c2.x); c2.y = fmaf(y.y, y.y, c2.y);
    c2.z = fmaf(y.z, y.z, c2.z); c2.w = fmaf(y.w, y.w, c2.w);
  }
  if (!a.euclid) return dot;
  float4 v;
  v.x = 2.f * dot.x; v.y = 2.f * dot.y; v.z = 2.f * dot.z; v.w = 2.f * dot.w;
  if (a.euclid == 2) return v;  // residual part1 = 2 q_j.r_jc (residual_part1_kernel)
  v.x = v.x - q2; v.y = v.y - q2; v.z = v.z - q2; v.w = v.w - q2;
  v.x = v.x - c2.x; v.y = v.y - c2.y; v.z = v.z - c2.z; v.w = v.w - c2.w;
  return v;
}

__device__ __forceinline__ void stage_lut_linear(const ScanArgs& a, int q, float* lut,
                                                 const float* xq) {
  // lut[j*256 + c] <- a.lut[(j*nq + q)*256 + c]; 16-byte loads, 1 KiB rows
  const float4* __restrict__ src = reinterpret_cast<const float4*>(a.lut);
  float4* dst = reinterpret_cast<float4*>(lut);
  for (int i = threadIdx.x; i < a.m * 64; i += kScanThreads) {
    const int j = i >> 6, c4 = i & 63;
    dst[i] = a.lut ? src[((int64_t)j * a.nq + q) * 64 + c4] : fused_lut4(a, j, c4, xq);
  }
}

// ---- reference-layout kernel ---------------------------------------------------------------

template <int R>
__global__ __launch_bounds__(kScanThreads) void scan_ref_kernel(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lut_bytes = a.m * 1024;
  const int list_bytes = kScanWaves * R * 64 * 8;
  const int region0 = lut_bytes > list_bytes ? lut_bytes : list_bytes;
  float* lut = reinterpret_cast<float*>(smem);
  float* qv_all = reinterpret_cast<float*>(smem + region0);
  int* qi_all = reinterpret_cast<int*>(smem + region0 + kScanWaves * 256);
  int* ptab = reinterpret_cast<int*>(smem + region0 + kScanWaves * 512);
  ProbeTable tab{ptab, ptab + a.max_nprobe, ptab + 2 * a.max_nprobe};
  unsigned* tau_key = reinterpret_cast<unsigned*>(ptab + 3 * a.max_nprobe + 1);
  float* xq = reinterpret_cast<float*>(tau_key + 1);  // fused LUT: query [m*ds], then |q_j|^2 [m]

  const int q = blockIdx.x / a.n_split;
  const int part = blockIdx.x - q * a.n_split;
  if (a.only_flagged && a.only_flagged[q] != a.epoch) return;  // exact redo of flagged queries only
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  int n_probe = (int)a.n_probe_list[q];
  n_probe = n_probe < 0 ? 0 : (n_probe > a.max_nprobe ? a.max_nprobe : n_probe);

  if (wave == 0) {
    build_probe_table(a, q, n_probe, tab);
    if (lane == 0) *tau_key = f2key(-INFINITY);
  }
  if (!a.lut) stage_query(a, q, xq, kScanThreads);
  stage_lut_linear(a, q, lut, xq);
  __syncthreads();

  WaveSelector<R> sel;
  sel.init(qv_all + wave * 64, qi_all + wave * 64, a.k);

  const int total_tiles = tab.tile_begin[n_probe];
  const int t_begin = (int)(((int64_t)total_tiles * part) / a.n_split);
  const int t_end = (int)(((int64_t)total_tiles * (part + 1)) / a.n_split);
  const int G = a.m >> 2;
  const uint32_t* __restrict__ codes32 = reinterpret_cast<const uint32_t*>(a.codes);

  int p = 0;
  for (int T = t_begin + wave; T < t_end; T += kScanWaves) {
    while (T >= tab.tile_begin[p + 1]) ++p;
    const int off = ((T - tab.tile_begin[p]) << 6) + lane;
    const bool valid = off < tab.size[p];
    const int s = tab.start[p] + off;
    float v = 0.f;
    bool live = valid;
    if (valid) {
      if (a.is_empty) live = (a.is_empty[s] == 0);  // ivfpq_topk.cu:878,883-884
      int g = 0;
      for (; g + 4 <= G; g += 4) {
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = codes32[(int64_t)(g + u) * a.n_slots + s];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* row = lut + (g + u) * 1024;
          v += row[w[u] & 255u];
          v += row[256 + ((w[u] >> 8) & 255u)];
          v += row[512 + ((w[u] >> 16) & 255u)];
          v += row[768 + (w[u] >> 24)];
        }
      }
      for (; g < G; ++g) {
        const uint32_t w = codes32[(int64_t)g * a.n_slots + s];
        const float* row = lut + g * 1024;
        v += row[w & 255u];
        v += row[256 + ((w >> 8) & 255u)];
        v += row[512 + ((w >> 16) & 255u)];
        v += row[768 + (w >> 24)];
      }
    }
    // workgroup-shared admission threshold: any wave's k-th best bounds the final k-th best
    const float tau_s = key2f(lds_poll_u32(tau_key));
    sel.tau = fmaxf(sel.tau, tau_s);
    const float tau_before = sel.tau;
    sel.push(live && (v >= sel.tau), v, s);
    if (sel.tau > tau_before && lane == 0) atomicMax(tau_key, f2key(sel.tau));
  }
  {
    const float tau_before = sel.tau;
    sel.flush();
    if (sel.tau > tau_before && lane == 0) atomicMax(tau_key, f2key(sel.tau));
  }
  finish_query<R>(a, q, part, sel.top, reinterpret_cast<float*>(smem),
                  reinterpret_cast<int*>(smem + kScanWaves * R * 64 * 4));
  // exact redo (one workgroup per query): the flag is consumed -- every thread read it before the first barrier.
  // (a captured graph replays with the same epoch: a flag left raised would redo the query on every replay)
  // Diagnostics: ws_delta[q] = kRedoneMark says "this query was redone exactly" -- on the routes whose
  // ws_delta holds a selection band the finisher's 1.f / 0.f never appears, so IVFPQTopkHip.last_redone reads this mark
  // (a band is >= 0; nobody reads ws_delta after this kernel, the call's last).
  if (a.only_flagged && threadIdx.x == 0) {
    const_cast<int*>(a.only_flagged)[q] = 0;
    if (a.ws_delta) a.ws_delta[q] = kRedoneMark;
  }
}

// ---- residual PQ (reference layout, exact) --------------------------------------------------
// Replaces ivfpq_topk_residual_precomputed (ivfpq_topk.cu:1039-1208) and ivfpq_topk_residual
// (:973-1037).  The LUT depends on the probed cell: LUT_p = part1[q] + part2[cell] (one fp32 add
// per entry, load_precomputed_v3 :522-560) or LUT_p = full[q][p]; value(slot) starts at
// base_sims[q][p] (:1113) and adds LUT_p[j][code_j] in ascending j.  One workgroup per query
// walks its probes in order: barrier, rebuild the 64-KiB LUT in LDS, barrier, the 8 waves scan
// the cell's tiles; the per-wave register top-k and shared threshold carry across cells.
struct ResidualArgs {
  const float* part1;       // [nq][m][256]        (mode A; nullptr = 2 q_j.r_jc built from query/codebook)
  const float* part2;       // [n_cells][m][256]   (mode A)
  const float* full;        // [nq][max_nprobe][m][256] (mode B) or nullptr
  const int64_t* cells;     // [nq][max_nprobe]    (mode A)
  const float* base_sims;   // [nq][max_nprobe]
  const float* slot_term;   // packed kernel: [n_slots] sum_j part2[cell(s)][j][code_j(s)]
  const float* cell_bound;  // packed kernel: [n_cells] sum_j max_c |part2[cell][j][c]|
};

template <int R>
__global__ __launch_bounds__(kScanThreads) void scan_residual_kernel(ScanArgs a, ResidualArgs ra) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lut_bytes = a.m * 1024;
  const int list_bytes = kScanWaves * R * 64 * 8;
  const int region0 = lut_bytes > list_bytes ? lut_bytes : list_bytes;
  float* lut = reinterpret_cast<float*>(smem);
  float* qv_all = reinterpret_cast<float*>(smem + region0);
  int* qi_all = reinterpret_cast<int*>(smem + region0 + kScanWaves * 256);
  int* ptab = reinterpret_cast<int*>(smem + region0 + kScanWaves * 512);
  ProbeTable tab{ptab, ptab + a.max_nprobe, ptab + 2 * a.max_nprobe};
  unsigned* tau_key = reinterpret_cast<unsigned*>(ptab + 3 * a.max_nprobe + 1);

  float* xq = reinterpret_cast<float*>(tau_key + 1);  // part1 built here: query [m*ds], |q_j|^2 [m]

  const int q = blockIdx.x;
  if (a.only_flagged && a.only_flagged[q] != a.epoch) return;  // exact redo of flagged queries only
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  int n_probe = (int)a.n_probe_list[q];
  n_probe = n_probe < 0 ? 0 : (n_probe > a.max_nprobe ? a.max_nprobe : n_probe);
  if (wave == 0) {
    build_probe_table(a, q, n_probe, tab);
    if (lane == 0) *tau_key = f2key(-INFINITY);
  }
  const bool build_part1 = !ra.full && !ra.part1;
  if (build_part1) stage_query(a, q, xq, kScanThreads);
  __syncthreads();

  WaveSelector<R> sel;
  sel.init(qv_all + wave * 64, qi_all + wave * 64, a.k);
  const int G = a.m >> 2;
  const uint32_t* __restrict__ codes32 = reinterpret_cast<const uint32_t*>(a.codes);
  const int n4 = a.m * 64;  // float4 count of one LUT

  for (int p = 0; p < n_probe; ++p) {
    const int sz = tab.size[p];
    if (sz == 0) continue;  // empty, or same start as the previous probe (ivfpq_topk.cu:1092-1107)
    __syncthreads();        // every wave is done with the previous cell's LUT
    float4* dst = reinterpret_cast<float4*>(lut);
    if (ra.full) {
      const float4* __restrict__ src =
          reinterpret_cast<const float4*>(ra.full) + ((int64_t)q * a.max_nprobe + p) * n4;
      for (int i = threadIdx.x; i < n4; i += kScanThreads) dst[i] = src[i];
    } else {
      const float4* __restrict__ s1 = reinterpret_cast<const float4*>(ra.part1) + (int64_t)q * n4;
      const float4* __restrict__ s2 = reinterpret_cast<const float4*>(ra.part2) +
                                      ra.cells[(int64_t)q * a.max_nprobe + p] * (int64_t)n4;
      for (int i = threadIdx.x; i < n4; i += kScanThreads) {
        const float4 x = build_part1 ? fused_lut4(a, i >> 6, i & 63, xq) : s1[i];
        const float4 y = s2[i];
        dst[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
      }
    }
    __syncthreads();
    const float base = ra.base_sims[(int64_t)q * a.max_nprobe + p];
    const int start = tab.start[p];
    const int tiles = (sz + 63) >> 6;
    for (int t = wave; t < tiles; t += kScanWaves) {
      const int off = (t << 6) + lane;
      const bool valid = off < sz;
      const int s = start + off;
      float v = base;
      bool live = valid;
      if (valid) {
        if (a.is_empty) live = (a.is_empty[s] == 0);
        for (int g = 0; g < G; ++g) {
          const uint32_t w = codes32[(int64_t)g * a.n_slots + s];
          const float* row = lut + g * 1024;
          v += row[w & 255u];
          v += row[256 + ((w >> 8) & 255u)];
          v += row[512 + ((w >> 16) & 255u)];
          v += row[768 + (w >> 24)];
        }
      }
      const float tau_s = key2f(lds_poll_u32(tau_key));
      sel.tau = fmaxf(sel.tau, tau_s);
      const float tau_before = sel.tau;
      sel.push(live && (v >= sel.tau), v + 0.0f, s);
      if (sel.tau > tau_before && lane == 0) atomicMax(tau_key, f2key(sel.tau));
    }
  }
  sel.flush();
  finish_query<R>(a, q, 0, sel.top, reinterpret_cast<float*>(smem),
                  reinterpret_cast<int*>(smem + kScanWaves * R * 64 * 4));
  if (a.only_flagged && threadIdx.x == 0) {  // consumed, and marked as redone (see scan_ref_kernel)
    const_cast<int*>(a.only_flagged)[q] = 0;
    if (a.ws_delta) a.ws_delta[q] = kRedoneMark;
  }
}

// ---- packed-layout kernel ------------------------------------------------------------------
// LUT in LDS in block order (scan_layout.h): entry (j, c) at dword lut_dword(M, j, c); the slot at
// address s stores at byte position p the code of sub-quantizer subq_at(M, p, s), so lane (slot s)
// step p reads a bank that differs from every other lane of its half-wave.
//
// The permuted order changes the fp32 summation order, so the streamed value f ("fast") is
// used for SELECTION only: with |f - e| <= delta (e = the reference's ascending-order value),
// every element of the exact top-k has f >= F_k - 2*delta (F_k = k-th best fast value).  Each
// wave keeps its best 64R > k candidates by f and admits everything down to threshold - 2*delta.
// At the end of the query a wave re-evaluates the entries that can still matter
// (f >= shared threshold - 2*delta: ~k/8 of them) exactly -- ascending j, from the packed bytes
// un-permuted through a private LDS row, LUT still resident -- re-ranks them by (e desc, address
// asc) and dumps the list; scan_merge_refine_kernel merges the per-wave lists of a query and
// writes the best k: bit-identical to the reference-layout kernel.  If a merged list ends up so
// full of near-ties (more than 64R candidates within 2*delta of the k-th) that a member of the
// exact top-k may have been evicted, the query is flagged and redone by scan_ref_kernel.

// jmax[j] (zeroed by the caller) collects max_c |LUT[j][c]| as the IEEE bit pattern of a
// non-negative float -- order-preserving as an unsigned, and LDS integer atomics are fast
// (float ones are not: DESIGN 3.5)
template <int M>
__device__ __forceinline__ void stage_lut_blocked(const ScanArgs& a, int q, float* lut,
                                                  int n_threads, unsigned* jmax, const float* xq,
                                                  const float* part1 = nullptr) {
  // thread handles (j, 4 consecutive c): 16-byte global load, 4 scalar LDS stores
  const float4* __restrict__ src = reinterpret_cast<const float4*>(a.lut);
  constexpr int m = M;
  // a wave-instruction covers JB sub-quantizers x 64/JB consecutive float4: 16 x 4 when m allows
  // (16 cache lines per load instead of one per lane, at the price of a 2-way bank conflict on the
  // stores: 1.2 % of the kernel in a same-box A/B; 8 x 8 and 32 x 2 measured slower), 8 x 8 or
  // 4 x 16 for m = 8 (mod 16) / 4 (mod 8)
  constexpr int JS = (m & 15) == 0 ? 4 : ((m & 7) == 0 ? 3 : 2);
  constexpr int JB = 1 << JS, CB = 64 >> JS;
  constexpr int jblocks = m >> JS;
  auto place = [&](int i, int& j, int& c4) {
    const int g = i >> 6, r = i & 63;
    const int jb = g % jblocks, cb4 = g / jblocks;
    j = jb * JB + (r & (JB - 1));
    c4 = cb4 * CB + (r >> JS);
  };
  auto put = [&](int j, int c4, const float4& x) {
    const int c = c4 * 4;
    lut[scan_layout::lut_dword(m, j, c + 0)] = x.x;
    lut[scan_layout::lut_dword(m, j, c + 1)] = x.y;
    lut[scan_layout::lut_dword(m, j, c + 2)] = x.z;
    lut[scan_layout::lut_dword(m, j, c + 3)] = x.w;
    const float mx = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
    atomicMax(&jmax[j], __float_as_uint(mx));
  };
  constexpr int DSM = M <= 32 ? 4 : 2;  // (m = 64 keeps 2 x 4 loads in flight: its eight-wave kernels sit at the VGPR cap)
  if (!part1 && !a.lut && a.ds <= DSM) {
    // fused table, short sub-vectors: a thread's entries come from ds codebook loads each, and a plain loop
    // pays one L2 round trip per entry group (8 groups per thread at m = 64: 4.6 of the 23 us a single-query
    // workgroup lives; 8.5 us when 512 workgroups stage at once).  All loads of U entry groups are issued first.
    // (round 6: ds = 3, 4 too -- SIFT's m = 32 walked its 8 groups per thread one round trip at a time: 14.2 of the
    // 15.5 us a workgroup of the reference grid's IVF4096 x 32 probes spent before its first tile)
    constexpr int U = TPQ_LUT_U;
    const int ds = a.ds;
    for (int i0 = threadIdx.x; i0 < m * 64; i0 += U * n_threads) {
      float4 y[U][DSM];
      int j[U], c4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * n_threads;
        place(i < m * 64 ? i : i0, j[u], c4[u]);
        const float4* __restrict__ cb = reinterpret_cast<const float4*>(a.codebook) + (int64_t)j[u] * ds * 64 + c4[u];
#pragma unroll
        for (int e = 0; e < DSM; ++e) y[u][e] = e < ds ? cb[e * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i0 + u * n_threads >= m * 64) break;
        float4 dot = make_float4(0.f, 0.f, 0.f, 0.f), c2 = dot;
        float q2 = 0.f;
#pragma unroll
        for (int e = 0; e < DSM; ++e) {
          if (e >= ds) break;
          const float4 yy = y[u][e];
          const float x = xq[j[u] * ds + e];
          q2 = fmaf(x, x, q2);
          dot.x = fmaf(x, yy.x, dot.x); dot.y = fmaf(x, yy.y, dot.y);
          dot.z = fmaf(x, yy.z, dot.z); dot.w = fmaf(x, yy.w, dot.w);
          c2.x = fmaf(yy.x, yy.x, c2.x); c2.y = fmaf(yy.y, yy.y, c2.y);
          c2.z = fmaf(yy.z, yy.z, c2.z); c2.w = fmaf(yy.w, yy.w, c2.w);
        }
        float4 v = dot;  // (fused_lut4's arithmetic, operation for operation)
        if (a.euclid) {
          v.x = 2.f * dot.x; v.y = 2.f * dot.y; v.z = 2.f * dot.z; v.w = 2.f * dot.w;
          if (a.euclid != 2) {
            v.x = v.x - q2; v.y = v.y - q2; v.z = v.z - q2; v.w = v.w - q2;
            v.x = v.x - c2.x; v.y = v.y - c2.y; v.z = v.z - c2.z; v.w = v.w - c2.w;
          }
        }
        put(j[u], c4[u], v);
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < m * 64; i += n_threads) {
    int j, c4;
    place(i, j, c4);
    const float4 x = part1 ? reinterpret_cast<const float4*>(part1)[((int64_t)q * m + j) * 64 + c4]
                     : a.lut ? src[((int64_t)j * a.nq + q) * 64 + c4]
                             : fused_lut4(a, j, c4, xq);
    put(j, c4, x);
  }
}

template <int M>
struct LdsLut {
  const float* lut;
  __device__ __forceinline__ float operator()(int j, unsigned c) const {
    return lut[scan_layout::lut_dword(M, j, (int)c)];
  }
};
// residual PQ: entry = part1 (LDS) + part2[cell] (global, L2-resident), rounded like the LUT the
// reference builds per probe (load_precomputed_v3, ivfpq_topk.cu:522-560)
template <int M>
struct ResidualLut {
  const float* lut;
  const float* part2_cell;  // this lane's cell: [M][256]
  __device__ __forceinline__ float operator()(int j, unsigned c) const {
    return lut[scan_layout::lut_dword(M, j, (int)c)] + part2_cell[j * 256 + (int)c];
  }
};
// Exact (ascending-j) value of slot `idx` from its PACKED bytes: the lane un-permutes its slot
// into sub-quantizer order through a private LDS row (stride M/4+1 dwords: conflict-free), then
// sums LUT entries in the reference's order.
template <int M, class LutFn>
__device__ __forceinline__ float exact_from_chunks(const typename scan_layout::Layout<M>::chunk_t (&w)[scan_layout::Layout<M>::kChunks],
                                                   int idx, bool active, uint32_t* scratch, int row_id,
                                                   const LutFn& lutfn, float init = 0.f) {
  using L = scan_layout::Layout<M>;
  constexpr int G = M / 4;
  uint32_t* row = scratch + row_id * (G + 1);
  if (active) {
#pragma unroll
    for (int d = 0; d < G; ++d) {
      const scan_layout::BlockAt<M> kb(4 * d);
      const int sb = idx & (kb.size - 1);
      const uint32_t x = (uint32_t)(sb & 3);
      const uint32_t sel = 0x03020100u ^ (x * 0x01010101u);  // out.byte[k] = in.byte[k ^ x]
      const uint32_t wd = L::word(w, d);
      const int dst = (kb.base >> 2) + ((d - (kb.base >> 2)) ^ (sb >> 2));
      row[dst] = __builtin_amdgcn_perm(wd, wd, sel);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  float v = init;
  if (active) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint32_t wd = row[g];
      v += lutfn(4 * g + 0, wd & 255u);
      v += lutfn(4 * g + 1, (wd >> 8) & 255u);
      v += lutfn(4 * g + 2, (wd >> 16) & 255u);
      v += lutfn(4 * g + 3, wd >> 24);
    }
  }
  return active ? v : -INFINITY;
}
template <int M, class LutFn>
__device__ __forceinline__ float exact_from_packed(const uint8_t* __restrict__ packed,
                                                   int64_t n_slots, int idx, bool active,
                                                   uint32_t* scratch, int row_id,
                                                   const LutFn& lutfn, float init = 0.f) {
  using L = scan_layout::Layout<M>;
  typename L::chunk_t w[L::kChunks] = {};
  if (active) L::load(packed, n_slots, idx, w);
  return exact_from_chunks<M>(w, idx, active, scratch, row_id, lutfn, init);
}

// The same value with EVERY lane evaluating a slot of its own and no LDS row (plain PQ, LUT in LDS).  The packed layout
// stores sub-quantizer j of slot s at byte position j ^ (s mod block) so that the scan's lanes read 64 different LUT
// rows at a time; 64 candidates summed in sub-quantizer order would all read the SAME row at a time (one bank, 64-way).
// So, sixteen sub-quantizers at a time: the lane picks the four code dwords that hold them (the XOR's high bits move
// whole groups of 16: a select among the block's groups), fetches their LUT entries in POSITION order -- the XOR's low
// four bits spread the lanes over 16 rows --, brings the VALUES (not the codes) into sub-quantizer order with a butterfly
// of conditional swaps on those four bits, and adds them ascending j: the reference's order, hence its bits.  One pass
// for 64 candidates where the LDS-row form took 64 / refine_rows passes of a 64-step dependent LDS chain each (~3 us a
// pass); sixteen values live at a time (all 64 at once spilled registers into the scan's tile loop).  Used by the pool
// mode's drains (64 candidates at a time).  NOT by the short lists' refinement: at k = 100 a wave has ~10 candidates and
// one LDS-row pass is the faster form; carrying both forms put 25 more scratch reloads into every query's finish of the
// k <= 248 kernels -- 3 % at C2, 8-15 % on the reference grid's short cells, same box (k = 500: +7 %).  (Not code size:
// the same kernels without their in-kernel redo, 53 -> 39 KB, run no faster.)
template <int M, int BASE>
__device__ __forceinline__ float exact_lane_blocks(const typename scan_layout::Layout<M>::chunk_t (&w)[scan_layout::Layout<M>::kChunks],
                                                   int idx, const float* __restrict__ lut, float v) {
  using L = scan_layout::Layout<M>;
  if constexpr (BASE >= M) {
    return v;
  } else {
    constexpr int S = scan_layout::block_of(M, BASE).size;
    constexpr int GS = S < 16 ? S : 16;   // sub-quantizers per group
    constexpr int NG = S / GS;            // groups in the block
    constexpr int DW = GS / 4;            // dwords per group
    const int xs = idx & (S - 1);
    const int xg = xs / GS, xl = xs & (GS - 1);
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      // sub-quantizers BASE + GS q ... + GS - 1 live at positions of group q ^ xg
      uint32_t cd[DW];
#pragma unroll
      for (int t = 0; t < DW; ++t) {
        cd[t] = L::word(w, BASE / 4 + (q ^ 0) * DW + t);
#pragma unroll
        for (int x = 1; x < NG; ++x) cd[t] = (xg == x) ? L::word(w, BASE / 4 + (q ^ x) * DW + t) : cd[t];
      }
      float val[GS];
#pragma unroll
      for (int p = 0; p < GS; ++p) {
        const uint32_t c = (cd[p >> 2] >> (8 * (p & 3))) & 255u;
        // position GS (q ^ xg) + p holds sub-quantizer BASE + GS q + (p ^ xl): lut_dword(M, that, c)
        val[p] = lut[BASE * 256 + (int)c * S + GS * q + (p ^ xl)];
      }
#pragma unroll
      for (int b = 1; b < GS; b <<= 1) {
        const bool sw = (xl & b) != 0;
#pragma unroll
        for (int p = 0; p < GS; ++p) {
          if ((p & b) == 0) {
            const float lo = val[p], hi = val[p | b];
            val[p] = sw ? hi : lo;
            val[p | b] = sw ? lo : hi;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < GS; ++j) v += val[j];
    }
    return exact_lane_blocks<M, BASE + S>(w, idx, lut, v);
  }
}
template <int M>
__device__ __forceinline__ float exact_lane(const typename scan_layout::Layout<M>::chunk_t (&w)[scan_layout::Layout<M>::kChunks],
                                            int idx, const float* __restrict__ lut) {
  return exact_lane_blocks<M, 0>(w, idx, lut, 0.f);
}

// ---- the 16-bit selection table ("sel16") -------------------------------------------------------------
// T[j][c] = round((LUT[j][c] + A_j) * inv), A_j = max_c |LUT[j][c]|, inv = 65535 / (2 max_j A_j): u16, laid out by
// scan_layout::lut16_halfword.  F(slot) = sum_j T[j][code_j] is an EXACT integer (< 2^24: carried as a float), and
// |F - (e_real + sum_j A_j) * inv| <= 0.51 m + 1 (per entry: the fp32 roundings of x * inv + (A_j * inv + 0.5), 0.008, and the
// rounding to an integer, 0.5; + 1 for the rounding of inv itself), so the selection band of the fp32 fast value
// carries over with delta = (0.51 m + 1) + (m - 1) u sum_j A_j * inv units.  Half the LDS of the fp32 table: four
// workgroups per CU at m = 64 instead of two.
// phase 1: the thread's M * 64 / NT float4 groups of entries (stage_lut_blocked's placement and, entry for entry, its
// arithmetic) into registers; the per-sub-quantizer maxima of |x| as BIT PATTERNS (NaN and Inf order above every
// finite value: the caller sees them in the maximum) into jmax
template <int M, int NT>
__device__ __forceinline__ void lut16_compute(const ScanArgs& a, int q, const float* xq, unsigned* jmax,
                                              float4 (&ent)[M * 64 / NT]) {
  constexpr int NE = M * 64 / NT;
  static_assert(M * 64 % NT == 0, "whole groups per thread");
  constexpr int JS = (M & 15) == 0 ? 4 : ((M & 7) == 0 ? 3 : 2);
  constexpr int JB = 1 << JS, CB = 64 >> JS;
  constexpr int jblocks = M >> JS;
  auto place = [&](int i, int& j, int& c4) {
    const int g = i >> 6, r = i & 63;
    const int jb = g % jblocks, cb4 = g / jblocks;
    j = jb * JB + (r & (JB - 1));
    c4 = cb4 * CB + (r >> JS);
  };
  auto note = [&](int j, const float4& x) {
    const unsigned b0 = __float_as_uint(x.x) & 0x7fffffffu, b1 = __float_as_uint(x.y) & 0x7fffffffu;
    const unsigned b2 = __float_as_uint(x.z) & 0x7fffffffu, b3 = __float_as_uint(x.w) & 0x7fffffffu;
    const unsigned m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3;
    atomicMax(&jmax[j], m01 > m23 ? m01 : m23);
  };
  const float4* __restrict__ src = reinterpret_cast<const float4*>(a.lut);
  if (!a.lut && a.ds <= 2) {
    constexpr int U = 4;
    static_assert(NE % U == 0, "batches of four");
    const int ds = a.ds;
#pragma unroll
    for (int u0 = 0; u0 < NE; u0 += U) {
      float4 y[U][2];
      int j[U], c4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        place((int)threadIdx.x + (u0 + u) * NT, j[u], c4[u]);
        const float4* __restrict__ cb = reinterpret_cast<const float4*>(a.codebook) + (int64_t)j[u] * ds * 64 + c4[u];
        y[u][0] = cb[0];
        y[u][1] = ds > 1 ? cb[64] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float4 dot = make_float4(0.f, 0.f, 0.f, 0.f), c2 = dot;
        float q2 = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (e >= ds) break;
          const float4 yy = y[u][e];
          const float x = xq[j[u] * ds + e];
          q2 = fmaf(x, x, q2);
          dot.x = fmaf(x, yy.x, dot.x); dot.y = fmaf(x, yy.y, dot.y);
          dot.z = fmaf(x, yy.z, dot.z); dot.w = fmaf(x, yy.w, dot.w);
          c2.x = fmaf(yy.x, yy.x, c2.x); c2.y = fmaf(yy.y, yy.y, c2.y);
          c2.z = fmaf(yy.z, yy.z, c2.z); c2.w = fmaf(yy.w, yy.w, c2.w);
        }
        float4 v = dot;  // (fused_lut4's arithmetic, operation for operation)
        if (a.euclid) {
          v.x = 2.f * dot.x; v.y = 2.f * dot.y; v.z = 2.f * dot.z; v.w = 2.f * dot.w;
          if (a.euclid != 2) {
            v.x = v.x - q2; v.y = v.y - q2; v.z = v.z - q2; v.w = v.w - q2;
            v.x = v.x - c2.x; v.y = v.y - c2.y; v.z = v.z - c2.z; v.w = v.w - c2.w;
          }
        }
        ent[u0 + u] = v;
        note(j[u], v);
      }
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    int j, c4;
    place((int)threadIdx.x + u * NT, j, c4);
    ent[u] = a.lut ? src[((int64_t)j * a.nq + q) * 64 + c4] : fused_lut4(a, j, c4, xq);
    note(j, ent[u]);
  }
}
// phase 2 (after a barrier: jmax is complete): quantise and store
template <int M, int NT>
__device__ __forceinline__ void lut16_store(const float4 (&ent)[M * 64 / NT], const unsigned* jmax, float inv,
                                            uint16_t* lut16) {
  constexpr int NE = M * 64 / NT;
  constexpr int JS = (M & 15) == 0 ? 4 : ((M & 7) == 0 ? 3 : 2);
  constexpr int JB = 1 << JS, CB = 64 >> JS;
  constexpr int jblocks = M >> JS;
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int i = (int)threadIdx.x + u * NT;
    const int g = i >> 6, r = i & 63;
    const int j = (g % jblocks) * JB + (r & (JB - 1));
    const int c = ((g / jblocks) * CB + (r >> JS)) * 4;
    // T = trunc(x * inv + (A_j * inv + 0.5)): ONE fma per entry (round 6; it was add, multiply, add, min -- the scan is
    // VALU-issue-bound).  |x| <= A_j, so the real value lies in [0.5, 65535.5]; roundings: the fma's (half an
    // ulp at < 2^16: 2^-8) and the constant's two (2^-9 each) -- the 0.008 the band's 0.51 per entry allows for; the
    // truncation of a value in (0.49, 65535.51) needs no clamp.
    const float k0 = __uint_as_float(jmax[j]) * inv + 0.5f;
    auto qz = [&](float x) -> uint16_t { return (uint16_t)(unsigned)fmaf(x, inv, k0); };
    lut16[scan_layout::lut16_halfword(M, j, c + 0)] = qz(ent[u].x);
    lut16[scan_layout::lut16_halfword(M, j, c + 1)] = qz(ent[u].y);
    lut16[scan_layout::lut16_halfword(M, j, c + 2)] = qz(ent[u].z);
    lut16[scan_layout::lut16_halfword(M, j, c + 3)] = qz(ent[u].w);
  }
}

// phase 2, wave-level: the merged list already carries EXACT values; write the best k and raise
// the overflow flag when the list is so full of near-ties that a member of the exact top-k may
// have been evicted from a wave's list (see the header comment of this section)
template <int R, bool RES = false>
__device__ __forceinline__ void finalize_and_write(const ScanArgs& a, int q, const WaveTopK<R>& top,
                                                   float delta2) {
  const float ek = top.kth_value(a.k);
  const Key klast = readlane_key(top.k[R - 1], 63);
  const bool overflow = (key_index(klast) != kPadIdx) && !(key_value(klast) < ek - delta2);
  write_final<R>(a, q, top);
  if (RES || a.small_lists) {  // the scan kernel may already have raised this one
    if (lane_id() == 0 && overflow) a.flags[q] = a.epoch;
  } else {
    if (lane_id() == 0) a.flags[q] = overflow ? a.epoch : 0;
  }
}

// Merge of L sorted lists (best first) of LEN keys each, lying in LDS as hi[l * LEN + i], lo[...], BY RANK:
// the position of an entry in the merged order is its own position plus, for every other list, the number of
// that list's entries that precede it -- a fixed-step binary search per list, eight lists' searches in
// flight per lane.  Equal keys (a slot scanned twice) rank by list: no two entries share a position.  Entries
// that land below `cap` are scattered into ohi / olo (pre-filled with pads by the caller); one barrier on
// either side instead of the 2 log2(L) of a tree of pairwise merges, and no serial chain of bitonic networks
// (the workgroup's 8 x 64: 4.6 -> 3.9 us, and 4 % of the C2 batch).
template <int LEN>
__device__ __forceinline__ void rank_merge(const unsigned* __restrict__ hi, const unsigned* __restrict__ lo, int L,
                                           unsigned* __restrict__ ohi, unsigned* __restrict__ olo, int cap, int tid,
                                           int n_threads) {
  static_assert((LEN & (LEN - 1)) == 0, "power of two");
  for (int e = tid; e < L * LEN; e += n_threads) {
    const int l = e / LEN;
    const Key x{hi[e], lo[e]};
    if (key_index(x) == kPadIdx) continue;
    const unsigned long long xu = key_u64(x);
    int rank = e - l * LEN;
    for (int l0 = 0; l0 < L; l0 += 8) {
      int cnt[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) cnt[j] = 0;
      // y precedes x: y > x, or y == x in an earlier list
      // (branch-free: a list beyond L or the entry's own list is searched like the others -- in bounds -- and
      // its count dropped; with a branch per list the eight searches ran one after the other, 90 cycles a read)
      int base[8];
      bool use[8], tie[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int l2 = l0 + j;
        use[j] = l2 < L && l2 != l;
        tie[j] = l2 < l;
        base[j] = (l2 < L ? l2 : L - 1) * LEN;
      }
#pragma unroll
      for (int s = LEN / 2; s >= 1; s >>= 1) {
        unsigned yh[8], yl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          yh[j] = hi[base[j] + cnt[j] + s - 1];
          yl[j] = lo[base[j] + cnt[j] + s - 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned long long yu = ((unsigned long long)yh[j] << 32) | yl[j];
          cnt[j] += (yu > xu || (yu == xu && tie[j])) ? s : 0;
        }
      }
      {
        unsigned yh[8], yl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          yh[j] = hi[base[j] + cnt[j]];
          yl[j] = lo[base[j] + cnt[j]];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned long long yu = ((unsigned long long)yh[j] << 32) | yl[j];
          const int c = cnt[j] + ((yu > xu || (yu == xu && tie[j])) ? 1 : 0);
          rank += use[j] ? c : 0;
        }
      }
    }
    if (rank < cap) {
      ohi[rank] = x.hi;
      olo[rank] = x.lo;
    }
  }
}

// waves per workgroup: 8 while two workgroups share a CU (LUT <= 64 KiB); 16 when the LUT is so
// large that only one workgroup fits (m > 64, e.g. GIST m=120: 120 KiB) -- same 16 waves per CU.
// Short codes (m <= 32, LUT <= 32 KiB): 4 waves, FOUR workgroups per CU -- a query is then a
// quarter of the CU's waves, so its fixed costs (launch, staging, end-of-query barrier, counting
// rounds, refinement: ~40 % of a query's life at m=16) overlap with three other queries' streaming
// instead of one (r02, 10 000 queries x 32 probes: m=8 0.98 -> 0.79 ms, 16 1.30 -> 1.12,
// 24 1.63 -> 1.51, 32 1.98 -> 1.73)
constexpr int packed_waves(int M) { return M <= 32 ? 4 : (M <= 64 ? 8 : 16); }
// Short codes are instruction-bound, not bandwidth-bound (~61 + 3.4 m cycles per 64-slot
// tile per CU, the 61 being table walk, address arithmetic, exec-mask handling, threshold poll and
// ballot): a lane therefore takes S slots (64 apart) per iteration and pays that part once.
// (measured, 10 000 queries x 32 probes: m=4 +18 %, 8 +16 %, 12 +12 %, 16 +9 %, 20 +10 %, 24 +10 %;
// neutral from m=28 on, where one slot per lane is kept)
#ifdef TPQ_SLOTS_LOG2  // experiments (tools/build_variant.sh): slots per lane = 1 << TPQ_SLOTS_LOG2
constexpr int packed_slots(int M) { return 1 << TPQ_SLOTS_LOG2; }
constexpr int packed_tile_shift(int M) { return 6 + TPQ_SLOTS_LOG2; }
#else
// (r02 sweep, 10 000 queries x 32 probes, ms for S = 1 / 2 / 4: m=28 1.97 / 2.06 / 2.04,
// m=32 2.25 / 2.05 / 1.98, m=40 2.37 / 2.44 / 2.47, m=48 2.84 / 2.65 / 4.82, m=56 3.35 / 3.21 / -,
// m=64 3.09 / 5.84 / -: the 16-byte-chunk layouts (m % 16 == 0) gain until the second tile's
// registers spill; with 4-wave workgroups (m <= 32): m=16 1.17 / 1.13 / 1.12, m=24 1.62 / 1.52 /
// 1.58, m=28 1.93 / 1.81 / 1.78, m=32 2.02 / 1.82 / 1.76)
// (round 6, after the look-ups of the small blocks went from 3.25 to 2 VALU: the per-tile part weighs more, and four
// slots per lane now win from m = 12 on -- same box, S = 2 -> 4, C2 shape k = 100 / k = 1 / 244-slot cells: m = 12 +5 / +6 /
// +2 %, 16 +7 / +9 / +7 %, 20 +8 / +6 / +4 %, 24 0 / +3 / +3 %; m = 40 -13 %, 48 -10 %, 56 -26 %: those keep theirs)
// (m = 40: two slots per lane once the per-slot part had shrunk -- +4 % at the C2 shape, +8 % on 244-slot cells, same box)
constexpr int packed_slots(int M) {
  return M <= 32 ? 4 : ((M == 40 || M == 48 || M == 56) ? 2 : 1);
}
constexpr int packed_tile_shift(int M) { return packed_slots(M) == 4 ? 8 : (packed_slots(M) == 2 ? 7 : 6); }
#endif

// per-wave scratch of the end-of-query exact re-evaluation: un-permute rows of M/4+1 dwords,
// 16 per pass (8 when the LUT leaves little LDS: m > 64)
constexpr int refine_rows(int M) { return M <= 64 ? 16 : 8; }
constexpr int packed_aux_bytes(int /*R*/, int M) {
  return packed_waves(M) * refine_rows(M) * (M / 4 + 1) * 4;
}

// 2 workgroups per CU (LDS: 2 x (64 KiB LUT + ~14 KiB)) need <= 128 VGPRs: 4 waves per SIMD.
// Long lists (R = 8, 16) are held to the same cap: a handful of spilled registers in the (cold)
// flush path cost far less than running one workgroup per CU (k = 300: 8.9 -> 6.4 ms).
//
// RES = residual PQ (replaces ivfpq_topk_residual_precomputed, ivfpq_topk.cu:1039-1208, at full
// scan speed): the reference rebuilds LUT_p = part1[q] + part2[cell_p] in shared memory for every
// probe (128 KiB read per 62 KiB of codes at C2).  Here only part1[q] is staged, once per query;
// the cell-dependent half of the fast value, sum_j part2[cell(s)][j][code_j(s)], is a per-SLOT
// constant precomputed at index-build time (ResidualArgs::slot_term, 4 B per slot) and
//   f(s) = sum_j part1[j][code_j] (permuted order) + (base_p + slot_term[s]).
// f is again a selection key only (|f - e| <= delta with the bound below); survivors are
// re-evaluated with the reference's arithmetic: v = base_p; v += fl(part1 + part2) ascending j.
//
// RM > 0 ("fused finish", small batches): the workgroup also FINISHES -- its waves' exact lists are
// tree-merged through LDS, a query split over several workgroups meets in the last one to arrive (a ticket
// per query), which writes the result; an overflowing candidate band is redone, exactly, by that same
// workgroup.  One launch instead of three (scan, scan_merge_refine_kernel, the flagged redo): at one query
// the two extra launches were 25 of 64 us.  RM = registers of the merged list (list_regs_packed(k)).
//
// RM < 0 ("pool mode", k > 504 -- k > 248 at m <= 32 --, plain PQ; scan.hip holds the rule): folding 64 candidates into a sorted list of k + 8 (or even 2k / NW)
// entries is what made large k slow -- at k = 1000 the tile loop ran 275 us per query against 106 at k = 100.
// Here the sorted per-wave list (R registers) only serves the ADMISSION THRESHOLD: it holds the wave's
// ceil(k / NW) best (bound (b) below needs no more), and every admitted candidate is also appended to an
// unsorted pool in the workspace.  Nothing is ever evicted from a pool, so at the end of the query the counting
// rounds run over the pools, the entries at or above the cut are compacted through the wave's queue, re-evaluated
// exactly and written back -- unsorted; scan_pool_merge_kernel ranks a query's ~k exact candidates in LDS.
// A pool that fills up flags the query for the exact kernel.
// RM <= kDumpF32 ("dump", large batches of plain PQ, k <= 248): the workgroup ENDS after the tile loop -- its waves store
// their lists of fast values and scan_finish_exact_kernel does the rest at full occupancy (the end of a query -- barrier,
// counting rounds, refinement, merge: 17 of the 43 us a 16-probe query of 244-slot cells lives -- held a 64-KiB-LDS
// workgroup slot idle).  RM = kDumpSel16: the table is the 16-bit one (above), four waves per workgroup.
constexpr int scan_waves(int M, int RM) { return RM == kDumpSel16 ? 4 : packed_waves(M); }
template <int R, int M, bool RES, int RM = 0>
__global__ __launch_bounds__(scan_waves(M, RM) * 64, 4) void scan_packed_kernel(ScanArgs a,
                                                                                      ResidualArgs ra,
                                                                                      float delta_rel) {
  using L = scan_layout::Layout<M>;
  constexpr bool DUMP = RM <= kDumpF32, SEL16 = is_sel16(RM), POOL = RM < 0 && !DUMP;
  static_assert(!(DUMP && RES), "dump mode serves plain PQ");
  constexpr int NW = scan_waves(M, RM);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int lut_bytes = SEL16 ? M * 512 : M * 1024;
  constexpr int aux_bytes = DUMP ? 0 : packed_aux_bytes(R, M);
  float* lut = reinterpret_cast<float*>(smem);
  uint32_t* scratch_all = reinterpret_cast<uint32_t*>(smem + lut_bytes);
  float* qv_all = reinterpret_cast<float*>(smem + lut_bytes + aux_bytes);
  int* qi_all = reinterpret_cast<int*>(smem + lut_bytes + aux_bytes + NW * 256);
  int* ptab = reinterpret_cast<int*>(smem + lut_bytes + aux_bytes + NW * 512);
  ProbeTable tab{ptab, ptab + a.max_nprobe, ptab + 2 * a.max_nprobe};
  unsigned* tau_key = reinterpret_cast<unsigned*>(ptab + 3 * a.max_nprobe + 1);
  int* tile_ctr = reinterpret_cast<int*>(tau_key + 1);  // m > 64: next tile to hand out
  float* red = reinterpret_cast<float*>(tile_ctr + 1);  // [2 NW] reduction scratch
  float* wave_q = red + 2 * NW;                // [NW] each wave's r-th best
  float* pbase = wave_q + NW;                  // RES: [max_nprobe] base_sims of the probe
  int* pcell = reinterpret_cast<int*>(pbase + (RES ? a.max_nprobe : 0));  // RES: [max_nprobe] cell
  float* xq = reinterpret_cast<float*>(pcell + (RES ? a.max_nprobe : 0));
  // (wave-uniform by construction: told to the compiler, so that the tile index, the probe cursor and their compares
  // live on the scalar unit instead of in VGPRs behind exec masks -- the scan is VALU-issue-bound)
  // (same box, caller-supplied table, C2 shape, TB/s without / with the hint: m = 16 4.60 / 4.71, 20 4.37 / 4.71,
  // 24 4.69 / 5.02, 32 5.88 / 5.90, 64 6.92 / 7.07)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  {
    // scan_layout's look-up address folds the table's LDS address into lane constants that are XORed with position bits
    // (accumulate, accumulate16): the table must start at a multiple of 128 bytes.  It does -- the dynamic allocation starts
    // at 0 as long as this kernel declares no static __shared__ --; a build that breaks that traps instead of mis-scanning.
    typedef const __attribute__((address_space(3))) char* lds_char_ptr;
    if (((uint32_t)(uintptr_t)(lds_char_ptr)smem & 127u) != 0u) __builtin_trap();
  }
  int q, part, parts;  // query, this workgroup's part of it, the parts it is dealt in
  if (DUMP && (int)blockIdx.x < a.unsplit) {  // (tail split, ScanArgs::unsplit: the leading queries are not split)
    q = (int)blockIdx.x;
    part = 0;
    parts = 1;
  } else {
    const int first = DUMP ? a.unsplit : 0;
    const int b = (int)blockIdx.x - first;
    q = first + b / a.n_split;
    part = b - (q - first) * a.n_split;
    parts = a.n_split;
  }
  TPQ_PROF(a, blockIdx.x, 0);
  int n_probe = (int)a.n_probe_list[q];
  n_probe = n_probe < 0 ? 0 : (n_probe > a.max_nprobe ? a.max_nprobe : n_probe);

  unsigned* jmax = reinterpret_cast<unsigned*>(qv_all);  // [M] (the queues are not live yet)
  if (threadIdx.x < M) jmax[threadIdx.x] = 0u;
  __syncthreads();
  // (wave 0 issues the loads of the probe table FIRST and builds the table after the LUT is staged: done up front,
  // its two dependent global round trips kept the other waves at the staging barrier for 1.7 us per query)
  ProbeRegs probes0{0, 0};
  if (wave == 0) {
    probes0 = fetch_probes(a, q, n_probe, 0);
    if (lane == 0) {
      *tau_key = f2key(-INFINITY);
      *tile_ctr = 0;
    }
    if (lane < NW) wave_q[lane] = -INFINITY;
  }
  TPQ_PROF(a, blockIdx.x, 1);
  const float* part1 = RES ? ra.part1 : nullptr;
  if (!a.lut && !part1) stage_query(a, q, xq, NW * 64);
  TPQ_PROF(a, blockIdx.x, 10);  // (dump modes: sub-phases of the prologue, slots 10 ... 14)
  [[maybe_unused]] float inv16 = 0.f;  // SEL16: table units per unit of value
  if constexpr (SEL16) {
    float4 ent[M * 64 / (NW * 64)];
    lut16_compute<M, NW * 64>(a, q, xq, jmax, ent);
    TPQ_PROF(a, blockIdx.x, 11);
    if (wave == 0) build_probe_table(a, q, n_probe, tab, packed_tile_shift(M), &probes0);
    __syncthreads();
    TPQ_PROF(a, blockIdx.x, 12);
    unsigned jb = 0u;
    float sum = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < M; j0 += 64) {
      if (j0 + lane < M) {
        jb = jmax[j0 + lane] > jb ? jmax[j0 + lane] : jb;
        sum += __uint_as_float(jmax[j0 + lane]);
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)jb, d, 64);
      jb = o > jb ? o : jb;
      sum += __shfl_xor(sum, d, 64);
    }
    const float J = __uint_as_float(jb);
    // a table that cannot be scaled (NaN / Inf entries, overflow of 2 J or of the bound, all zeros): the exact kernel
    // takes the query (scan.hip launches it over the flagged ones)
    const bool scalable = jb < 0x7f800000u && J >= 1e-30f && J <= 1e37f && sum <= 1e37f;
    if (threadIdx.x == 0 && part == 0) a.flags[q] = scalable ? 0 : a.epoch;
    if (!scalable) return;  // (workgroup-uniform: every wave reduced the same words)
    inv16 = 65535.f / (2.f * J);
    lut16_store<M, NW * 64>(ent, jmax, inv16, reinterpret_cast<uint16_t*>(lut));
    TPQ_PROF(a, blockIdx.x, 13);
  } else {
    stage_lut_blocked<M>(a, q, lut, NW * 64, jmax, xq, part1);
    TPQ_PROF(a, blockIdx.x, 11);
    if (wave == 0) build_probe_table(a, q, n_probe, tab, packed_tile_shift(M), &probes0);
    TPQ_PROF(a, blockIdx.x, 13);
  }
  float probe_mx = 0.f;
  if constexpr (RES) {
    for (int pp = threadIdx.x; pp < n_probe; pp += NW * 64) {
      const float b = ra.base_sims[(int64_t)q * a.max_nprobe + pp];
      const int c = (int)ra.cells[(int64_t)q * a.max_nprobe + pp];
      pbase[pp] = b;
      pcell[pp] = c;
      probe_mx = fmaxf(probe_mx, fabsf(b) + ra.cell_bound[c]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) probe_mx = fmaxf(probe_mx, __shfl_xor(probe_mx, d, 64));
    if (lane == 0) red[NW + wave] = probe_mx;
  }
  __syncthreads();
  TPQ_PROF(a, blockIdx.x, 2);

  // delta >= |fast - exact|: both are fp32 sums of the same M terms in different orders, each
  // within (M-1) u * sum|x_i| of the real sum (u = 2^-24), and sum|x_i| <= sum_j max_c|LUT[j][c]|.
  // RES: the terms are base_p, part1_j, part2_j: exact = M sequential adds of fl(part1_j+part2_j)
  // onto base_p, fast = (M-1)-add sums of the part1's and of the part2's plus two more adds: each
  // within (M+1) u A of the real sum, A = |base_p| + sum_j max|part1_j| + cell_bound[cell_p]
  // (the host passes delta_rel with M+1 in place of M-1).
  // (sum_j max_c|LUT[j][c]| from the maxima collected while staging; every wave reduces the same
  // M words in the same order, so all of them hold the identical bound)
  float bound = 0.f;
#pragma unroll
  for (int j0 = 0; j0 < M; j0 += 64)
    if (j0 + lane < M) bound += __uint_as_float(jmax[j0 + lane]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) bound += __shfl_xor(bound, d, 64);
  __syncthreads();  // jmax lives in the queue area: everyone has read it before the first push
  if constexpr (RES) {
    float mx = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) mx = fmaxf(mx, red[NW + w]);
    bound += mx;
  }
  float delta2 = 2.f * delta_rel * bound;  // 2*delta: width of the candidate band
  if constexpr (SEL16) {
    // in table units: the quantisation (0.51 per entry, + 1) and the exact value's own distance from the real sum
    // ((M - 1) u bound = delta_rel bound / 2.1, taken as delta_rel bound / 2)
    delta2 = ceilf(2.f * (0.51f * (float)M + 1.f + 0.5f * delta_rel * bound * inv16)) + 1.f;
  } else if constexpr (DUMP) {
    // (a bound that is not finite: leave the query to the exact kernel, as the 16-bit table does)
    const bool ok = bound <= 1e37f;
    if (threadIdx.x == 0 && part == 0) a.flags[q] = ok ? 0 : a.epoch;
    if (!ok) return;
  }
  TPQ_PROF(a, blockIdx.x, 3);

  WaveSelector<R> sel;
  sel.init(qv_all + wave * 64, qi_all + wave * 64, a.k);
  sel.margin = delta2;
  typename WaveSelector<R>::Pool pool{nullptr, nullptr, 0, 0};
  if constexpr (POOL) {
    const int64_t o = (((int64_t)q * a.n_split + part) * NW + wave) * a.pool_cap;
    pool = {a.pool_hi + o, a.pool_lo + o, 0, a.pool_cap};
  }

  const int total_tiles = tab.tile_begin[n_probe];
  const int t_begin = (int)(((int64_t)total_tiles * part) / parts);
  const int t_end = (int)(((int64_t)total_tiles * (part + 1)) / parts);

  // Workgroup-shared admission threshold.  Two valid lower bounds of the final k-th best:
  //  (a) any wave's own k-th best (tau_key, atomic max);
  //  (b) min over the 8 waves of each wave's r-th best, r = ceil(k/8): the 8 lists then hold
  //      >= 8r >= k candidates at or above it.  Tiles are dealt round-robin to the waves, so
  //      (b) tracks the true k-th best closely and keeps the pass rate near k*ln(N/k)/N.
  const int r_share = (a.k + NW - 1) / NW;
  // readers poll ONE word per tile; the (rare) publisher folds bound (b) into it
  auto refresh_tau = [&]() {
    sel.tau = fmaxf(sel.tau, key2f(lds_poll_u32(tau_key)));
  };
  auto publish = [&](float /*tau_before*/) {
    // readlane must run with every lane active: inside `if (lane == 0)` the source lane is
    // inactive and its register contents are undefined to the compiler
    const float mine = sel.top.kth_value(r_share);
    if (lane == 0) {
      lds_post_f32(wave_q + wave, mine);
      float qmin = lds_poll_f32(wave_q);
#pragma unroll
      for (int w = 1; w < NW; ++w) qmin = fminf(qmin, lds_poll_f32(wave_q + w));
      atomicMax(tau_key, f2key(fmaxf(sel.tau, qmin)));
    }
  };

  constexpr bool kOneAhead = R > 4 || (R >= 2 && NW == 8 && !DUMP);  // (tile loop of m <= 64: see there)
  if constexpr (packed_slots(M) == 1) {
    constexpr int kFetchLoads = L::kChunks + (RES ? 1 : 0);  // global loads of one fetch (without tombstones)
    struct Tile {
      int s;
      bool valid;
      float add;      // RES: base_p + slot_term[s]
      uint32_t lim;   // the cell's last slot (a wave past its last tile: slot 0)
    };
    int p = 0;
    // Every global load of the tile loop is UNCONDITIONAL, on a clamped address (round 6).  With the prefetch under
    // `if (next tile exists) if (lane has a slot)` hipcc's waitcnt bookkeeping merged the two paths at the join and
    // put `s_waitcnt vmcnt(3 .. 0)` in front of the four chunks of the CURRENT tile's look-ups -- i.e. the wave waited for
    // the first chunks of the tile it had just prefetched before consuming the tile already in its registers (the ISA
    // of the loop: eight loads in flight wanted vmcnt(7 .. 4)).  A lane without a slot, and the whole wave past its
    // last tile, read the LAST slot of the tile's cell instead (one v_min_u32 against a wave-uniform bound -- was compare +
    // select of slot 0; the line is one the live lanes touch anyway; a wave past its last tile: slot 0, a tile exists, so
    // slot 0 does) and drop the value.
    auto locate = [&](int T) -> Tile {
      while (T >= tab.tile_begin[p + 1]) ++p;
      const int off = ((T - tab.tile_begin[p]) << 6) + lane;
      const int st = tab.start[p], sz = tab.size[p];
      Tile t{st + off, off < sz, 0.f, (uint32_t)(st + sz - 1)};
      return t;
    };
    auto fetch = [&](int T, Tile& t, typename L::chunk_t (&w)[L::kChunks]) {
      if (T < t_end) {  // (wave-uniform; nothing is loaded from global memory inside)
        t = locate(T);
      } else {
        t.valid = false;
        t.lim = 0u;
      }
      const uint32_t s = min((uint32_t)t.s, t.lim);
      L::load_u(a.packed, a.n_slots, s, w);
      if constexpr (RES) t.add = (T < t_end ? pbase[p] : 0.f) + ra.slot_term[s];
    };
    // (a lane without a slot carries NaN: it fails the admission compare by itself -- no `live` flag is kept in a register
    // next to the value; the scan is VALU-issue-bound.  Tombstones -- a foreign index with holes inside its cells,
    // ivfpq_topk.cu:878,883-884 -- are looked up for the candidates that PASS the threshold only (round 6; the flag bytes
    // used to travel with the prefetch: one more load per slot in flight, a number of loads per fetch that depended on the
    // call, and a fetch whose loads hipcc's waitcnt bookkeeping could not count exactly).)
    auto consume = [&](const typename L::chunk_t(&w)[L::kChunks], const Tile& t) {
      float v = __builtin_nanf("");
      if (t.valid) {
        if constexpr (SEL16) v = (float)L::accumulate16(w, t.s, reinterpret_cast<const uint16_t*>(lut));
        else v = L::accumulate(w, t.s, lut);
        if constexpr (RES) v += t.add;
      }
      // Every load of THIS tile has landed on every path past this point -- said explicitly (round 6): a wave whose tile has
      // no live lane branches around the look-ups and their `s_waitcnt vmcnt(7 .. 4)`, hipcc's waitcnt bookkeeping merged
      // that path in at the loop header, saw a load pending on the registers the next fetch reuses as temporaries and put
      // `s_waitcnt vmcnt(0)` in front of every other prefetch: the wave drained its loads before issuing the next tile's.
      // What remains in flight here is the prefetched tile (one fetch = kFetchLoads loads; with tombstones one more per
      // slot: that call waits for the first of them too).
      if constexpr (M <= 64) wait_vmcnt<kFetchLoads>();
      refresh_tau();
      const float tau_before = sel.tau;
      const int flushes_before = sel.n_flush;
      bool pass = v >= sel.tau - delta2;
      if (a.is_empty) {  // (wave-uniform)
        if (pass) pass = a.is_empty[t.s] == 0;
      }
      if constexpr (POOL) sel.push_pool(pool, pass, v, t.s);
      else sel.push(pass, v, t.s);
      if (sel.n_flush != flushes_before) publish(tau_before);
    };

    // software pipeline: the codes of tile T+NW are in flight while tile T is being consumed
    // (m <= 64; larger m runs 16 waves per workgroup under a 128-VGPR cap and relies on them)
    if constexpr (M <= 64) {
      typename L::chunk_t w0[L::kChunks], w1[L::kChunks];
      Tile m0{0, false, 0.f, 0u}, m1{0, false, 0.f, 0u};
      // TWO tiles ahead (round 6): a register set is refilled -- with the tile after next -- right behind its own
      // look-ups, so one to two tiles of loads are in flight at every moment and the probe-table walk of a fetch uses the
      // registers of the tile just consumed as its temporaries: m = 16 +5 %, 24 +4..13 %, 48 +4..7 % on the same box.
      // (hipcc re-rotates the loop and still puts `s_waitcnt vmcnt(0)` in front of every other prefetch -- DESIGN 3.1 --,
      // so the wave does drain once per two tiles; what the order buys is the earlier issue of the other prefetch.)
      // (Lists of two registers and more in the eight-wave workgroups of the sorted-list path -- m > 32, k = 300 / 500 --
      // keep the one-ahead order: two ahead cost them 4-5 % on the same box.)
      int T = t_begin + wave;
      if constexpr (kOneAhead) {
        if (T < t_end) fetch(T, m0, w0);
        while (T < t_end) {
          fetch(T + NW, m1, w1);
          consume(w0, m0);
          T += NW;
          if (T >= t_end) break;
          fetch(T + NW, m0, w0);
          consume(w1, m1);
          T += NW;
        }
      } else if (T < t_end) {  // (a wave without a tile loads nothing: slot 0 need not exist)
        fetch(T, m0, w0);
        fetch(T + NW, m1, w1);
#ifdef TPQ_SCAN_PROFILE
        bool first_tile = true;
#endif
        while (true) {
          consume(w0, m0);
#ifdef TPQ_SCAN_PROFILE
          if (first_tile) TPQ_PROF(a, blockIdx.x, 14);
          first_tile = false;
#endif
          fetch(T + 2 * NW, m0, w0);
          T += NW;
          if (T >= t_end) break;
          consume(w1, m1);
          fetch(T + 2 * NW, m1, w1);
          T += NW;
          if (T >= t_end) break;
        }
      }
    } else {
      // One 16-wave workgroup per CU and one tile in flight per wave: with a static deal the waves
      // drift apart (the oldest wave of a SIMD wins the issue arbitration), the early finishers
      // idle at the end-of-query barrier and the stragglers run alone, latency-bound -- 37-41 % of
      // the workgroup's life at m = 120.  Tiles are therefore handed out from an LDS counter (one
      // integer atomic per tile, fetched while the previous tile is consumed); a wave's tile
      // indices still increase, which is all locate() needs.
      auto grab = [&]() -> int {
        int t = 0;
        if (lane == 0) t = atomicAdd(tile_ctr, 1);
        return t_begin + __builtin_amdgcn_readfirstlane(t);
      };
      typename L::chunk_t w0[L::kChunks];
      Tile m0{0, false, 0.f, 0u};
      int T = grab();
      while (T < t_end) {
        fetch(T, m0, w0);
        const int Tn = grab();
        consume(w0, m0);
        T = Tn;
      }
    }
  } else {
    constexpr int S = packed_slots(M);          // slots per lane per tile, 64 apart
    constexpr int TS = packed_tile_shift(M);    // log2(slots per tile)
    constexpr int kFetchLoads = S * (L::kChunks + (RES ? 1 : 0));  // global loads of one fetch (without tombstones)
    struct Tile {
      int s;      // the lane's first slot; its u-th slot is s + 64 u
      int rem;    // slots of the cell from s on: the u-th slot exists iff 64 u < rem
      float add;  // RES: base_p (slot_term is added per slot)
      uint32_t lim;  // the cell's last slot (a wave past its last tile: slot 0)
    };
    int p = 0;
    auto locate = [&](int T) -> Tile {
      while (T >= tab.tile_begin[p + 1]) ++p;
      const int off = ((T - tab.tile_begin[p]) << TS) + lane;
      const int st = tab.start[p], sz = tab.size[p];
      Tile t{st + off, sz - off, 0.f, (uint32_t)(st + sz - 1)};
      if constexpr (RES) t.add = pbase[p];
      return t;
    };
    // (every global load unconditional, on a clamped address: see the one-slot-per-lane loop above)
    struct Side {
      float term[S];     // RES: slot_term of the lane's slots
    };
    auto fetch = [&](int T, Tile& t, typename L::chunk_t (&w)[S][L::kChunks], Side& sd) {
      if (T < t_end) {  // (wave-uniform; nothing is loaded from global memory inside)
        t = locate(T);
      } else {
        t.rem = 0;
        t.lim = 0u;
      }
  #pragma unroll
      for (int u = 0; u < S; ++u) {
        // (a slot past the end of the cell: the cell's last slot; a wave past its last tile: slot 0 -- read and dropped)
        const uint32_t su = min((uint32_t)(t.s + 64 * u), t.lim);
        L::load_u(a.packed, a.n_slots, su, w[u]);
        if constexpr (RES) sd.term[u] = ra.slot_term[su];
      }
    };
    auto consume = [&](const typename L::chunk_t (&w)[S][L::kChunks], const Side& sd, const Tile& t) {
      // (a lane's missing slot carries NaN: it fails the admission compare by itself; tombstones are looked up for the
      // passing candidates only: see the one-slot-per-lane loop above)
      float v[S];
  #pragma unroll
      for (int u = 0; u < S; ++u) {
        v[u] = __builtin_nanf("");
        if (64 * u < t.rem) {
          if constexpr (SEL16) v[u] = (float)L::accumulate16(w[u], t.s + 64 * u, reinterpret_cast<const uint16_t*>(lut));
          else v[u] = L::accumulate(w[u], t.s + 64 * u, lut);
          if constexpr (RES) v[u] += t.add + sd.term[u];
        }
      }
      // (this tile's loads have landed on every path: see the one-slot-per-lane loop above)
      if constexpr (M <= 64) wait_vmcnt<kFetchLoads>();
      refresh_tau();
      if constexpr (S > 1) {
        bool any = false;
  #pragma unroll
        for (int u = 0; u < S; ++u) any = any || (v[u] >= sel.tau - delta2);
        if (__ballot(any) == 0ull) return;  // the common case: one ballot for S x 64 slots
      }
  #pragma unroll
      for (int u = 0; u < S; ++u) {
        const float tau_before = sel.tau;
        const int flushes_before = sel.n_flush;
        bool pass = v[u] >= sel.tau - delta2;
        if (a.is_empty) {  // (wave-uniform)
          if (pass) pass = a.is_empty[t.s + 64 * u] == 0;
        }
        if constexpr (POOL) sel.push_pool(pool, pass, v[u], t.s + 64 * u);
        else sel.push(pass, v[u], t.s + 64 * u);
        if (sel.n_flush != flushes_before) {
          publish(tau_before);
          // (the tile's remaining slots meet the threshold the flush just raised -- the first tiles of a query admit
          // everything, and a short list, 32 probes of 244 slots at k = 100, spends as much on its flushes as on its look-ups)
          refresh_tau();
        }
      }
    };

    // software pipeline: the codes of tile T+NW are in flight while tile T is being consumed
    // (m <= 64; larger m runs 16 waves per workgroup under a 128-VGPR cap and relies on them)
    if constexpr (M <= 64) {
      typename L::chunk_t w0[S][L::kChunks], w1[S][L::kChunks];
      Side r0 = {}, r1 = {};
      Tile m0{0, 0, 0.f, 0u}, m1{0, 0, 0.f, 0u};
      // (two tiles ahead, one ahead for long lists: see the one-slot-per-lane loop above)
      int T = t_begin + wave;
      if constexpr (kOneAhead) {
        if (T < t_end) fetch(T, m0, w0, r0);
        while (T < t_end) {
          fetch(T + NW, m1, w1, r1);
          consume(w0, r0, m0);
          T += NW;
          if (T >= t_end) break;
          fetch(T + NW, m0, w0, r0);
          consume(w1, r1, m1);
          T += NW;
        }
      } else if (T < t_end) {  // (a wave without a tile loads nothing: slot 0 need not exist)
        fetch(T, m0, w0, r0);
        fetch(T + NW, m1, w1, r1);
        while (true) {
          consume(w0, r0, m0);
          fetch(T + 2 * NW, m0, w0, r0);
          T += NW;
          if (T >= t_end) break;
          consume(w1, r1, m1);
          fetch(T + 2 * NW, m1, w1, r1);
          T += NW;
          if (T >= t_end) break;
        }
      }
    } else {
      // One 16-wave workgroup per CU and one tile in flight per wave: with a static deal the waves
      // drift apart (the oldest wave of a SIMD wins the issue arbitration), the early finishers
      // idle at the end-of-query barrier and the stragglers run alone, latency-bound -- 37-41 % of
      // the workgroup's life at m = 120.  Tiles are therefore handed out from an LDS counter (one
      // integer atomic per tile, fetched while the previous tile is consumed); a wave's tile
      // indices still increase, which is all locate() needs.
      auto grab = [&]() -> int {
        int t = 0;
        if (lane == 0) t = atomicAdd(tile_ctr, 1);
        return t_begin + __builtin_amdgcn_readfirstlane(t);
      };
      typename L::chunk_t w0[S][L::kChunks];
      Side r0 = {};
      Tile m0{0, 0, 0.f, 0u};
      int T = grab();
      while (T < t_end) {
        fetch(T, m0, w0, r0);
        const int Tn = grab();
        consume(w0, r0, m0);
        T = Tn;
      }
    }
  }
  TPQ_PROF(a, blockIdx.x, 4);
  {
    const float tau_before = sel.tau;
    sel.flush();
    publish(tau_before);
  }
  TPQ_PROF(a, blockIdx.x, 5);

  if constexpr (DUMP) {
    // ---- dump mode: the wave's list of fast values, whether it may have lost one, the band -- and out ----
    // (a flush folds at most 64 candidates in: a list of 64 R entries that has seen no more than R flushes evicted nothing)
    const int64_t li = ((int64_t)q * a.n_split + part) * NW + wave;
    store_list<R>(sel.top, a.ws_vals + li * (R * 64), a.ws_idx + li * (R * 64));
    if (lane == 0) a.list_evict[li] = sel.n_flush > R ? 1 : 0;
    if (part == 0 && wave == 0 && lane == 0) a.ws_delta[q] = delta2;
    TPQ_PROF(a, blockIdx.x, 6);
    return;
  } else if constexpr (POOL) {
    // ---- pool mode: cut, compaction, exact values ----
    static_assert(!RES, "pool mode serves plain PQ");
    __syncthreads();  // every wave has published its quantile
    TPQ_PROF(a, blockIdx.x, 6);
    float shared_tau;
    {
      float qmin = lds_poll_f32(wave_q);
#pragma unroll
      for (int w = 1; w < NW; ++w) qmin = fminf(qmin, lds_poll_f32(wave_q + w));
      shared_tau = fmaxf(qmin, key2f(lds_poll_u32(tau_key)));
    }
    constexpr int PR = RM == -1 ? 16 : 32;  // pool registers: pool_cap = 64 PR entries (1024 / 2048)
    const bool overflow = pool.n > pool.cap;
    const int n_use = overflow ? 0 : pool.n;
    unsigned ph[PR], pl[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      ph[r] = 0u;
      pl[r] = 0u;
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      if (r * 64 >= n_use) break;  // wave-uniform
      const int e = r * 64 + lane;
      const bool valid = e < n_use;
      // (agent-scope loads: the wave reads back what it stored itself, past its L1)
      ph[r] = valid ? __hip_atomic_load(pool.hi + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      pl[r] = valid ? __hip_atomic_load(pool.lo + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    // counting rounds over the pools: invariant "at least k pool entries of the workgroup are >= lo"
    unsigned lo = f2key(shared_tau), hi = 0xFFFFFFFFu;
    {
      unsigned* cnt = reinterpret_cast<unsigned*>(qv_all);  // [3][NW][NW] (the queues are empty)
      auto count_ge = [&](unsigned t) -> unsigned {
        unsigned c = 0;
#pragma unroll
        for (int r = 0; r < PR; ++r) {
          if (r * 64 >= n_use) break;  // wave-uniform
          c += (unsigned)__popcll(__ballot(ph[r] >= t && ph[r] != 0u));
        }
        return c;
      };
      auto wave_max = [&](unsigned x) -> unsigned {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          const unsigned o = (unsigned)__shfl_xor((int)x, d, 64);
          x = o > x ? o : x;
        }
        return x;
      };
#pragma unroll 1
      for (int round = 0; round < (RM == -3 ? 0 : 3); ++round) {
        unsigned my_t = 0;
        if (round == 0) {
          if (lane < NW) my_t = f2key(lds_poll_f32(wave_q + lane));
        } else {
          const unsigned long long span = (unsigned long long)(hi - lo);
          my_t = lo + (unsigned)((span * (unsigned)(lane + 1)) / (unsigned)(NW + 1));
        }
        unsigned mine = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const unsigned c = count_ge((unsigned)__builtin_amdgcn_readlane((int)my_t, j));
          mine = (lane == j) ? c : mine;
        }
        unsigned* cr = cnt + (round % 2) * NW * NW;
        if (lane < NW) cr[wave * NW + lane] = mine;
        __syncthreads();
        unsigned total = 0;
        if (lane < NW) {
#pragma unroll
          for (int w = 0; w < NW; ++w) total += cr[w * NW + lane];
        }
        const bool in = lane < NW;
        const bool ok = in && total >= (unsigned)a.k;
        const unsigned best_ok = wave_max(ok ? my_t : 0u);
        const unsigned best_no = ~wave_max((in && !ok) ? ~my_t : 0u);
        lo = best_ok > lo ? best_ok : lo;
        hi = best_no < hi ? best_no : hi;
        if (hi == 0xFFFFFFFFu || hi <= lo) break;  // workgroup-uniform
      }
    }
    __syncthreads();  // the counts lay over the queues
    TPQ_PROF(a, blockIdx.x, 7);
    const float cut = fmaxf(shared_tau, key2f(lo)) - delta2;
    constexpr int RR = refine_rows(M);
    constexpr int RX = NW == 4 ? 8 : 4;  // the wave's exact candidates, sorted: ~2 ceil(k / NW) entries at k = 1000
    uint32_t* scratch = scratch_all + wave * RR * (M / 4 + 1);
    int* qi = qi_all + wave * 64;
    int qn = 0, n_out = 0;
    WaveTopK<RX> ex;
    ex.init();
    auto drain = [&]() {  // the (<= 64) queued addresses: exact values, folded into the wave's sorted list
      if (qn == 0) return;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      const bool act = lane < qn;
      const int idx = act ? qi[lane] : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      typename L::chunk_t cw[L::kChunks] = {};
      if (act) L::load(a.packed, a.n_slots, idx, cw);
      const float e = exact_lane<M>(cw, idx, lut);
      ex.insert_unsorted(act ? make_key(e + 0.0f, idx) : pad_key());
      n_out += qn;
      qn = 0;
    };
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      if (r * 64 >= n_use) break;  // wave-uniform
      const bool want = ph[r] != 0u && key2f(ph[r]) >= cut;
      const unsigned long long wmask = __ballot(want);
      if (wmask == 0ull) continue;
      const int n = __popcll(wmask);
      if (qn + n > 64) drain();
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wmask, 0));
      if (want) qi[qn + rank] = (int)~pl[r];
      qn += n;
    }
    drain();
    // the sorted list goes where the pool was (its entries are all in registers by now); more candidates than
    // the list holds, or a pool that filled up: the exact kernel redoes the query
    store_list<RX>(ex, reinterpret_cast<float*>(pool.hi), reinterpret_cast<int*>(pool.lo));
    if (lane == 0 && (overflow || n_out > 64 * RX)) a.flags[q] = a.epoch;
    TPQ_PROF(a, blockIdx.x, 8);
    TPQ_PROF(a, blockIdx.x, 9);
    return;
  } else {

  // End of query, per wave and without any barrier: re-evaluate the surviving candidates of
  // this wave's list exactly (ascending j, LUT still in LDS), re-rank them by exact value and
  // dump the list; scan_merge_refine_kernel (one wave per query) merges the 8 x n_split lists.
  // Only entries that can still reach the top-k (f >= shared threshold - 2*delta) are touched:
  // with the quantile-shared threshold that is ~k/8 per wave, i.e. one 16-lane pass.
  {
    // One barrier: every wave has folded its last queue in and published its r-th best, so the
    // shared bound (b) is now computed from FRESH lists.  During the scan the lists lag (a wave
    // admits only ~k*ln(N/k)/NW candidates in its whole life and folds them in 64 at a time), so
    // the running threshold leaves ~100 entries per wave above it; the fresh bound leaves ~2k/NW.
    __syncthreads();
    TPQ_PROF(a, blockIdx.x, 6);
    float shared_tau;  // identical in every wave (the loop below must be workgroup-uniform)
    {
      float qmin = lds_poll_f32(wave_q);
#pragma unroll
      for (int w = 1; w < NW; ++w) qmin = fminf(qmin, lds_poll_f32(wave_q + w));
      shared_tau = fmaxf(qmin, key2f(lds_poll_u32(tau_key)));
      sel.tau = fmaxf(sel.tau, shared_tau);
    }
    // Two counting rounds pull the bound up to (nearly) the exact k-th best fast value of the
    // workgroup: invariant "at least k list entries are >= lo".  Round 0 tests the NW published
    // quantiles themselves, round 1 NW keys evenly spaced inside the bracket round 0 leaves; every
    // wave counts its own sorted registers (ballots), the NW x NW counts meet in the dead queue
    // area, and each wave reduces them redundantly -- two barriers, no list ever leaves registers.
    // Every candidate kept beyond the k-th costs an exact re-evaluation (M gathers; M cache lines
    // of the part2 table in the residual kernel), so the tight cut pays for itself.
    {
      unsigned* cnt = reinterpret_cast<unsigned*>(qv_all);  // [2][NW][NW]
      auto count_ge = [&](unsigned t) -> unsigned {
        unsigned c = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) c += (unsigned)__popcll(__ballot(sel.top.k[r].hi >= t));
        return c;
      };
      auto wave_max = [&](unsigned x) -> unsigned {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          const unsigned o = (unsigned)__shfl_xor((int)x, d, 64);
          x = o > x ? o : x;
        }
        return x;
      };
      unsigned lo = f2key(shared_tau), hi = 0xFFFFFFFFu;
#pragma unroll 1
      for (int round = 0; round < 2; ++round) {
        unsigned my_t = 0;  // lane j < NW: threshold j of this round
        if (round == 0) {
          if (lane < NW) my_t = f2key(lds_poll_f32(wave_q + lane));
        } else {
          const unsigned long long span = (unsigned long long)(hi - lo);
          my_t = lo + (unsigned)((span * (unsigned)(lane + 1)) / (unsigned)(NW + 1));
        }
        unsigned mine = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const unsigned c = count_ge((unsigned)__builtin_amdgcn_readlane((int)my_t, j));
          mine = (lane == j) ? c : mine;
        }
        unsigned* cr = cnt + round * NW * NW;
        if (lane < NW) cr[wave * NW + lane] = mine;
        __syncthreads();
        unsigned total = 0;
        if (lane < NW) {
#pragma unroll
          for (int w = 0; w < NW; ++w) total += cr[w * NW + lane];
        }
        const bool in = lane < NW;
        const bool ok = in && total >= (unsigned)a.k;
        const unsigned best_ok = wave_max(ok ? my_t : 0u);           // largest threshold still >= k
        const unsigned best_no = ~wave_max((in && !ok) ? ~my_t : 0u);  // smallest one below k
        lo = best_ok > lo ? best_ok : lo;
        hi = best_no < hi ? best_no : hi;
        if (hi == 0xFFFFFFFFu || hi <= lo) break;  // wave-uniform: nothing left to bracket
      }
      sel.tau = fmaxf(sel.tau, key2f(lo));
    }
    TPQ_PROF(a, blockIdx.x, 7);
    const float cut = sel.tau - delta2;
    if (a.small_lists) {
      // Large k: the per-wave lists hold 64R < k + 8 entries (tiles are dealt round-robin, so a
      // wave's share of the top-k is ~k/NW; R is sized for twice that).  A wave whose list is FULL
      // of candidates that can still matter may have evicted one that matters too: flag the query
      // for the exact kernel.  (A list whose worst entry is below the cut lost nothing: everything
      // it evicted was worse still.)
      // (A flush folds at most 64 candidates in, so a list of 64 R entries that has seen no more than R
      // flushes evicted nothing at all: a query of a few hundred slots -- n_probe 1 or 2 on the reference's
      // benchmark grid -- fills lists whose cut is still -inf, and must not take the exact redo for it.)
      const Key kl = readlane_key(sel.top.k[R - 1], 63);
#ifndef TPQ_EXP_NO_OVERFLOW_FLAG  // knock-out for tests/test_gpu_kernels.py's adversarial case
      // (write-through, agent scope: with the fused finish the reader is the query's LAST workgroup, possibly on
      // another XCD, inside this launch -- a plain store could still sit in this XCD's L2 when it looks)
      if (sel.n_flush > R && key_index(kl) != kPadIdx && key_value(kl) >= cut && lane == 0)
        __hip_atomic_store(a.flags + q, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    constexpr int RR = refine_rows(M);
    uint32_t* scratch = scratch_all + wave * RR * (M / 4 + 1);
    WaveTopK<R> ex;
    ex.init();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int idx = key_index(sel.top.k[r]);
      const bool want = (idx != kPadIdx) && (key_value(sel.top.k[r]) >= cut);
      const unsigned long long wmask = __ballot(want);
      if (wmask == 0ull) break;  // sorted by fast value: nothing further down qualifies either
      float e = -INFINITY;
      float init = 0.f;
      const float* p2 = ra.part2;
      if constexpr (RES) {
        // which probe does the candidate's slot belong to?  (first match in probe order; a slot
        // covered by two probes -- a cell listed twice, non-adjacent -- is scanned twice by the
        // reference with two different bases: leave such queries to the exact kernel)
        int myp = -1, n_match = 0;
        for (int pp = 0; pp < n_probe; ++pp) {
          const bool hit = want && ((unsigned)(idx - tab.start[pp]) < (unsigned)tab.size[pp]);
          myp = (hit && myp < 0) ? pp : myp;
          n_match += hit ? 1 : 0;
        }
        if (n_match > 1) a.flags[q] = a.epoch;
        myp = myp < 0 ? 0 : myp;
        init = pbase[myp];
        p2 = ra.part2 + (int64_t)pcell[myp] * (M * 256);
      }
      // every wanted lane fetches its candidate's packed bytes NOW, in one batch: loaded inside the passes
      // below (16 rows each: the un-permute scratch is 16 rows per wave), each pass waited out a memory
      // latency of its own -- 6.5 of the 45 us a 16-probe query of 244-slot cells lives at k = 100
      typename L::chunk_t cw[L::kChunks] = {};
      if (want) L::load(a.packed, a.n_slots, idx, cw);
#pragma unroll 1
      for (int pass = 0; pass < 64 / RR; ++pass) {
        if (((wmask >> (RR * pass)) & ((1ull << RR) - 1ull)) == 0ull) continue;  // wave-uniform
        const bool mine = want && ((lane / RR) == pass);
        float ep;
        if constexpr (RES)
          ep = exact_from_chunks<M>(cw, idx, mine, scratch, lane % RR, ResidualLut<M>{lut, p2}, init);
        else
          ep = exact_from_chunks<M>(cw, idx, mine, scratch, lane % RR, LdsLut<M>{lut});
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        e = mine ? ep : e;
      }
      ex.insert_unsorted(want ? make_key(e, idx) : pad_key());
    }
    TPQ_PROF(a, blockIdx.x, 8);
    if constexpr (RM > 0 && !RES) {
      static_assert(RM >= R, "merged list shorter than the per-wave lists");
      // ---- fused finish ----
      WaveTopK<RM> mt;
      mt.init();
#pragma unroll
      for (int r = 0; r < R; ++r) mt.k[r] = ex.k[r];  // (sorted; the pads of init() rank last)
      float* lv = reinterpret_cast<float*>(smem);     // [NW][RM 64] x 2: over the LUT, the rows and the queues
      int* li = reinterpret_cast<int*>(smem + (size_t)NW * RM * 64 * 4);
      int* s_flag = tile_ctr;                         // (dead: m > 64 hands tiles out of it during the scan only)
      auto tree = [&]() {  // NW lists -> wave 0
        for (int stride = 1; stride < NW; stride <<= 1) {
          if ((wave & (2 * stride - 1)) == stride) store_list<RM>(mt, lv + wave * RM * 64, li + wave * RM * 64);
          __syncthreads();
          if ((wave & (2 * stride - 1)) == 0)
            merge_list<RM>(mt, lv + (wave + stride) * RM * 64, li + (wave + stride) * RM * 64);
          __syncthreads();
        }
      };
      // (merge area: over the LUT, the rows and the queues -- everything below the probe table: fuse_fits())
      unsigned* mhi = reinterpret_cast<unsigned*>(smem);
      auto load_merged = [&](const unsigned* ohi, const unsigned* olo) {
#pragma unroll
        for (int r = 0; r < RM; ++r) mt.k[r] = Key{ohi[r * 64 + lane], olo[r * 64 + lane]};
      };
      __syncthreads();  // every wave is done with the LUT and its rows
      {  // the workgroup's NW lists -> one, by rank (rank_merge)
        constexpr int LEN = 64 * R;
        unsigned* mlo = mhi + NW * LEN;
        unsigned* ohi = mlo + NW * LEN;
        unsigned* olo = ohi + RM * 64;
        store_list<R>(ex, reinterpret_cast<float*>(mhi + wave * LEN), reinterpret_cast<int*>(mlo + wave * LEN));
        const Key pad = pad_key();
        for (int i = threadIdx.x; i < RM * 64; i += NW * 64) {
          ohi[i] = pad.hi;
          olo[i] = pad.lo;
        }
        __syncthreads();
        rank_merge<LEN>(mhi, mlo, NW, ohi, olo, RM * 64, (int)threadIdx.x, NW * 64);
        __syncthreads();
        if (wave == 0) load_merged(ohi, olo);
      }
      TPQ_PROF(a, blockIdx.x, 9);
      bool last = true;
      if (a.n_split > 1) {
        // the workgroup's list -> workspace; release; ticket.  (G16 of the CDNA guide: plain stores, wait,
        // agent-scope release by one lane, relaxed agent-scope ticket; the last arriver acquires)
        if (wave == 0) {
          const int64_t o = ((int64_t)q * a.n_split + part) * (RM * 64);
          // (write-through stores -- relaxed, agent scope: sc1 -- need no cache write-back before the ticket)
#pragma unroll
          for (int r = 0; r < RM; ++r) {
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.ws_vals) + o + r * 64 + lane, mt.k[r].hi, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.ws_idx) + o + r * 64 + lane, mt.k[r].lo, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) {
            const int t = __hip_atomic_fetch_add(a.tickets + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int is_last = t == a.n_split - 1;
            if (is_last) {
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
              a.tickets[q] = 0;  // (zero on exit: the next call's workgroups start from it)
            }
            *s_flag = is_last;
          }
        }
        __syncthreads();
        last = *s_flag != 0;
        TPQ_PROF(a, blockIdx.x, 10);
        if (last) {  // block-uniform
          // (plain loads: the acquire above invalidated this CU's view; the lists were written back by their
          // producers' releases)
          // (wave w folds the lists of splits w, w + NW, ...; then the tree.  Ranking 16 x 128 entries against
          // each other in LDS, as the workgroup's own lists are merged above, measured 26 us against 7)
          mt.init();
          for (int pp = wave; pp < a.n_split; pp += NW) {
            const int64_t o = ((int64_t)q * a.n_split + pp) * (RM * 64);
            merge_list<RM>(mt, a.ws_vals + o, a.ws_idx + o);
          }
          __syncthreads();
          tree();
        }
      } else {
        TPQ_PROF(a, blockIdx.x, 10);
      }
      if (!last) return;
      TPQ_PROF(a, blockIdx.x, 11);
      // wave 0 holds the query's list, exact values: write, and decide whether the band overflowed
      if (wave == 0) {
        const float ek = mt.kth_value(a.k);
        const Key klast = readlane_key(mt.k[RM - 1], 63);
        bool overflow = (key_index(klast) != kPadIdx) && !(key_value(klast) < ek - delta2);
        if (a.small_lists) overflow = overflow || (__hip_atomic_load(a.flags + q, __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT) == a.epoch);
        write_final<RM>(a, q, mt);
        if (lane == 0) {
          // the flag is consumed here: a graph replays with the SAME epoch, and a flag left raised would send
          // every later replay of this query through the redo (diagnostics: ws_delta[q] = 1 when it was redone)
          a.flags[q] = 0;
          a.ws_delta[q] = overflow ? 1.f : 0.f;
          *s_flag = overflow;
        }
      }
      __syncthreads();
      TPQ_PROF(a, blockIdx.x, 12);
      if (*s_flag == 0) return;
      // ---- the exact redo (normally never): this workgroup rescans the query's probed cells with the
      // reference's arithmetic (ascending j, from the packed bytes) and overwrites the result ----
      __syncthreads();
      if (threadIdx.x < M) jmax[threadIdx.x] = 0u;
      __syncthreads();
      stage_lut_blocked<M>(a, q, lut, NW * 64, jmax, xq, nullptr);  // (the merge buffers lay over it)
      __syncthreads();
      if (wave == 0 && lane == 0) *tau_key = f2key(-INFINITY);
      __syncthreads();
      WaveSelector<RM> xs;
      xs.init(qv_all + wave * 64, qi_all + wave * 64, a.k);
      for (int pp = 0; pp < n_probe; ++pp) {
        const int size = tab.size[pp], start = tab.start[pp];
        for (int off0 = wave * 64; off0 < size; off0 += NW * 64) {
          const int off = off0 + lane;
          const bool valid = off < size;
          const int sidx = start + (valid ? off : 0);
          float e = -INFINITY;
#pragma unroll 1
          for (int pass = 0; pass < 64 / RR; ++pass) {
            const bool mine = valid && ((lane / RR) == pass);
            const float ep = exact_from_packed<M>(a.packed, a.n_slots, sidx, mine, scratch, lane % RR, LdsLut<M>{lut});
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            e = mine ? ep : e;
          }
          bool live = valid;
          if (valid && a.is_empty) live = (a.is_empty[sidx] == 0);
          xs.tau = fmaxf(xs.tau, key2f(lds_poll_u32(tau_key)));
          const float tau_before = xs.tau;
          xs.push(live && (e >= xs.tau), e, sidx);
          if (xs.tau > tau_before && lane == 0) atomicMax(tau_key, f2key(xs.tau));
        }
      }
      xs.flush();
      mt = xs.top;
      __syncthreads();  // every wave is done with the LUT
      tree();
      if (wave == 0) write_final<RM>(a, q, mt);
      return;
    }
    const int64_t o = (((int64_t)q * a.n_split + part) * NW + wave) * (R * 64);
    store_list<R>(ex, a.ws_vals + o, a.ws_idx + o);
    TPQ_PROF(a, blockIdx.x, 9);
    if (part == 0 && wave == 0 && lane == 0) a.ws_delta[q] = delta2;
  }
  }  // (neither dump nor pool mode)
}

// ---- split merge ---------------------------------------------------------------------------

template <int R>
__global__ __launch_bounds__(64) void scan_merge_kernel(ScanArgs a) {
  const int q = blockIdx.x;
  if (a.only_flagged && a.only_flagged[q] != a.epoch) return;
  WaveTopK<R> top;
  top.init();
  for (int part = 0; part < a.n_split; ++part) {
    const int64_t o = ((int64_t)q * a.n_split + part) * (R * 64);
    merge_list<R>(top, a.ws_vals + o, a.ws_idx + o);
  }
  write_final<R>(a, q, top);
}

// packed path, phase 2: merge the per-wave lists of a query (exact values) and write the result.
// W = blockDim.x / 64 waves per query (host: min(8, n_lists / 2)): wave w folds lists w, w+W, ...
// rank-major (every list's best 64 first: once those are in, most later chunks fail the
// wave-uniform early-exit test of insert_sorted) with the loads issued a group ahead of the
// merges, then the W partial lists are tree-merged through LDS.  Small batches run with many
// splits per query (512 lists at nq = 1): one wave folding them serially took 0.2 ms.
// RL = registers per dumped list (64 RL entries each), R = registers of the merged result.
template <int RL, int R, int M, bool RES>
__global__ __launch_bounds__(512) void scan_merge_refine_kernel(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int q = blockIdx.x;
  const int lane = lane_id();
  const int W = (int)(blockDim.x >> 6), wave = (int)(threadIdx.x >> 6);
  const int n_lists = a.n_split * packed_waves(M);  // a multiple of 8
  const int n_mine = n_lists / W;
  WaveTopK<R> top;
  top.init();
  const unsigned* __restrict__ bv =
      reinterpret_cast<const unsigned*>(a.ws_vals) + (int64_t)q * n_lists * (RL * 64);
  const unsigned* __restrict__ bi =
      reinterpret_cast<const unsigned*>(a.ws_idx) + (int64_t)q * n_lists * (RL * 64);
  const int T = n_mine * RL;  // item t: rank chunk t / n_mine of my (t % n_mine)-th list
  auto load_item = [&](int t) -> Key {
    if (t >= T) return pad_key();
    const int r = t / n_mine, l = (t - r * n_mine) * W + wave;
    const int64_t o = (int64_t)l * (RL * 64) + r * 64 + lane;
    return Key{bv[o], bi[o]};
  };
  constexpr int G = 4;
  Key k0[G], k1[G];
#pragma unroll
  for (int u = 0; u < G; ++u) k0[u] = load_item(u);
  for (int t = 0; t < T; t += 2 * G) {
#pragma unroll
    for (int u = 0; u < G; ++u) k1[u] = load_item(t + G + u);
#pragma unroll
    for (int u = 0; u < G; ++u) top.insert_sorted(k0[u]);
#pragma unroll
    for (int u = 0; u < G; ++u) k0[u] = load_item(t + 2 * G + u);
#pragma unroll
    for (int u = 0; u < G; ++u) top.insert_sorted(k1[u]);
  }
  float* lv = reinterpret_cast<float*>(smem);
  int* li = reinterpret_cast<int*>(smem + (size_t)W * R * 64 * 4);
  for (int stride = 1; stride < W; stride <<= 1) {
    if ((wave & (2 * stride - 1)) == stride) store_list<R>(top, lv + wave * R * 64, li + wave * R * 64);
    __syncthreads();
    if ((wave & (2 * stride - 1)) == 0)
      merge_list<R>(top, lv + (wave + stride) * R * 64, li + (wave + stride) * R * 64);
    __syncthreads();
  }
  if (wave == 0) finalize_and_write<R, RES>(a, q, top, a.ws_delta[q]);
}

// dump modes, phase 2: ONE WAVE per query.  The query's lists of FAST values arrive as NCH chunks of 64 keys (n_split x
// nw_scan lists of RL chunks, best first).  With F_k the k-th best fast value over all of them and `band` the scan's
// 2 delta, every member of the exact top-k has F >= F_k - band.  F_k comes from a bit-wise binary search on the key
// images (a ballot and a scalar popcount per chunk and step: the chunks never leave their registers, and most of the
// work rides on the scalar unit); nothing at or above the cut may have been lost on the way -- a wave's list that evicted
// (list_evict) and still ends at or above the cut, or more survivors than 64 RM, flags the query for the exact kernel.
// The survivors are compacted through a small LDS queue, 64 per pass, and evaluated EXACTLY, one per lane: the
// candidate's 64 packed bytes are brought into sub-quantizer order IN REGISTERS (a byte permute per dword for the low
// two bits of its XOR mask, four rounds of conditional dword swaps for the others), then sub-quantizer by sub-quantizer,
// the same j in every lane: entry = tpq_adc_lut's arithmetic on the codebook row (in LDS) and the query component
// (v_readlane from a register: wave-uniform), added in ascending j -- the reference's order, hence its bits.  Sorted by
// (value desc, address asc) and written.
// The codebook lives in LDS: one persistent workgroup per CU copies it (m * ds KiB, query-independent) once and its
// waves walk the queries.  (Entries fetched from global memory -- a different cache line per lane and look-up -- ran into
// the address coalescer: 396 us per 10 000 queries at k = 100; from LDS with lane-varying sub-quantizers and a value
// butterfly: 160 us, instruction-bound at ~6 000 VALU per query; this form: ~2 500.)  Nothing of a scan workgroup's
// table slot is held while this runs, which is the point of the split: the end of a query idled that slot for 17 of
// its 43 us.
// (waves per workgroup: 16 at every RM the kernel is built for -- RM = 8, k in (248, 504], ds = 2: 128 KiB of codebook + 32
// KiB of survivor queues, all of the CU's LDS; a longer exact list would halve them)
//
// Round 6: every packed block structure (the 64-block of m = 64 and the 32 / 16 / 8 / 4-blocks of the shorter codes:
// the un-permute below walks scan_layout's blocks), any sub-vector length with m * ds <= 128 (DS = 0: read from the
// arguments), and a second SOURCE of the exact entries -- FROM_LUT: the caller's materialised table [m][nq][256]
// (tpq_adc_lut's output, the reference boundary: IVFPQTopkCuda.topk(precomputed=...), kernels/IVFPQTopkCuda.py:81-142),
// gathered per survivor (m independent loads per lane, ascending-j adds); nothing is staged in LDS then.
constexpr int finish_waves(int RM) { return RM <= 8 ? 16 : 8; }
// registers of the finish kernel's exact list.  Round 6: 16 (eight waves per workgroup: 128 KiB of codebook + 32 KiB of
// survivor queues) -- k in (440, 504] on long cells, whose band holds more than the 512 candidates of RM = 8
constexpr int kDumpMaxR = 16;
static size_t finish_lds_bytes(int m, int ds, int RM, bool from_lut) {
  return (from_lut ? 0 : (size_t)m * ds * 1024) + (size_t)finish_waves(RM) * RM * 64 * 4;
}
template <int RM, int M, int DS, int NCH, bool FROM_LUT = false>
__global__ __launch_bounds__(finish_waves(RM) * 64) void scan_finish_exact_kernel(ScanArgs a, int nw_scan, int RL) {
  constexpr int kFinishWaves = finish_waves(RM);
  using L = scan_layout::Layout<M>;
  constexpr int G = M / 4;  // code dwords per slot
  static_assert(DS == 0 || M * DS <= 128, "the query rides in two registers per lane");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int ds = DS ? DS : a.ds;  // (the host admits m * ds <= 128 only)
  float* cb = reinterpret_cast<float*>(smem);  // [m][ds][256]
  if constexpr (!FROM_LUT) {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a.codebook);
    float4* dst = reinterpret_cast<float4*>(cb);
    for (int i = threadIdx.x; i < M * ds * 64; i += kFinishWaves * 64) dst[i] = src[i];
  }
  int* qi = reinterpret_cast<int*>(cb + (FROM_LUT ? 0 : M * ds * 256)) + wave * (RM * 64);  // the wave's survivors (addresses)
  __syncthreads();
  const int n_lists_all = a.n_split * nw_scan;
  const int T_all = n_lists_all * RL;  // chunks per query in the workspace (<= NCH)
  const bool euclid = a.euclid != 0;
  for (int q = (int)blockIdx.x * kFinishWaves + wave; q < a.nq; q += (int)gridDim.x * kFinishWaves) {
    if (a.flags[q] == a.epoch) continue;  // the scan left the query to the exact kernel
    // (tail split: an unsplit query filled the lists of its one part only; the stride is that of n_split parts)
    const int n_lists = q < a.unsplit ? nw_scan : n_lists_all;
    const int T = n_lists * RL;  // chunks in use
    // the query: component i in lane i % 64 of register i / 64; |q_j|^2 (ascending-dimension fma chain) in lane j
    float xv[2] = {0.f, 0.f}, q2v = 0.f;
    if constexpr (!FROM_LUT) {
      if (lane < M * ds) xv[0] = a.query[(int64_t)lane * a.nq + q];
      if (64 + lane < M * ds) xv[1] = a.query[(int64_t)(64 + lane) * a.nq + q];
      if (lane < M) {
        for (int e = 0; e < ds; ++e) {
          const float x = a.query[(int64_t)(lane * ds + e) * a.nq + q];
          q2v = fmaf(x, x, q2v);
        }
      }
    }
    const unsigned* __restrict__ bv = reinterpret_cast<const unsigned*>(a.ws_vals) + (int64_t)q * T_all * 64;
    const unsigned* __restrict__ bi = reinterpret_cast<const unsigned*>(a.ws_idx) + (int64_t)q * T_all * 64;
    unsigned hi[NCH];
    int ix[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
      hi[t] = t < T ? bv[t * 64 + lane] : 0u;  // (0 < the image of -inf: never counted, never wanted)
      ix[t] = t < T ? (int)~bi[t * 64 + lane] : kPadIdx;
      if (ix[t] == kPadIdx) hi[t] = 0u;
    }
    int evict = 0;
    if (lane < n_lists) evict = a.list_evict[(int64_t)q * n_lists_all + lane];
    const float band = a.ws_delta[q];
    // F_k: the largest key image t with at least k entries >= t (0 while fewer than k entries exist)
    unsigned fk = 0u;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned t = fk | (1u << bit);
      int n = 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) n += __popcll(__ballot(hi[c] >= t));
      fk = n >= a.k ? t : fk;
    }
    const float cut = (fk ? key2f(fk) : -INFINITY) - band;
    const unsigned cutk = f2key(cut);
    // survivors -> queue; a list that evicted and still ends at or above the cut lost one that mattered
    int n_c = 0;
    bool lost = false;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const bool want = hi[c] != 0u && hi[c] >= cutk;
      const unsigned long long mask = __ballot(want);
      const int n = __popcll(mask);
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
      if (want && n_c + rank < RM * 64) qi[n_c + rank] = ix[c];
      n_c += n;
      // (chunk c is rank chunk c % RL of list c / RL: its lane 63 is the list's last entry when c % RL == RL - 1)
      const int l = c / (RL > 0 ? RL : 1);
      const bool last_chunk = (c % (RL > 0 ? RL : 1)) == RL - 1;
      const int ev = __builtin_amdgcn_readlane(evict, l < 64 ? l : 0);
      lost = lost || (last_chunk && ev && ((mask >> 63) & 1ull));
    }
    lost = lost || n_c > RM * 64;
    if (lost) {  // (wave-uniform)
      if (lane == 0) a.flags[q] = a.epoch;
      continue;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    WaveTopK<RM> ex;
    ex.init();
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      if (r * 64 >= n_c) break;  // wave-uniform
      const bool want = r * 64 + lane < n_c;
      const int idx = want ? qi[r * 64 + lane] : 0;  // (idle lanes walk slot 0's bytes: in range)
      typename L::chunk_t cw[L::kChunks];
      L::load(a.packed, a.n_slots, idx, cw);
      // sub-quantizer order, block by block (scan_layout::subq_at): inside a block of B positions from base b,
      // out dword D byte Y = in dword D ^ (x >> 2), byte Y ^ (x & 3), x = idx mod B -- a byte permute per dword for
      // the low two bits of x, log2(B / 4) rounds of conditional dword swaps for the others
      unsigned cd[G];
#pragma unroll
      for (int d = 0; d < G; ++d) {
        constexpr int dummy = 0;
        (void)dummy;
        const scan_layout::BlockAt<M> kb(4 * d);
        const unsigned x = (unsigned)idx & (unsigned)(kb.size - 1);
        const unsigned sel = 0x03020100u ^ ((x & 3u) * 0x01010101u);
        const unsigned wd = L::word(cw, d);
        cd[d] = __builtin_amdgcn_perm(wd, wd, sel);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int d = 0; d < G; ++d) {
          const scan_layout::BlockAt<M> kb(4 * d);
          const int dr = d - (kb.base >> 2);            // dword inside the block
          if ((4 << b) < kb.size && (dr & (1 << b)) == 0) {  // the block has this XOR bit; d is the pair's lower dword
            const bool sw = (((unsigned)idx >> (2 + b)) & 1u) != 0u;  // (bit 2 + b of idx mod B: 4 << b < B)
            const unsigned lo = cd[d], up = cd[d | (1 << b)];         // (blocks are aligned to their size: | == +)
            cd[d] = sw ? up : lo;
            cd[d | (1 << b)] = sw ? lo : up;
          }
        }
      }
      float v = 0.f;
      if constexpr (FROM_LUT) {
        // the caller's table: entry (j, c) of query q at lut[(j * nq + q) * 256 + c]; all loads first, adds ascending j
        float ent[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const unsigned c = (cd[j >> 2] >> (8 * (j & 3))) & 255u;
          ent[j] = a.lut[((int64_t)j * a.nq + q) * 256 + (int)c];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) v += ent[j];
      } else {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const unsigned c = (cd[j >> 2] >> (8 * (j & 3))) & 255u;
          float dot = 0.f, c2 = 0.f;
          if constexpr (DS != 0) {
#pragma unroll
            for (int e = 0; e < DS; ++e) {
              const int i = j * DS + e;
              const float y = cb[i * 256 + (int)c];
              const float xx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[i >> 6]), i & 63));
              dot = fmaf(xx, y, dot);
              c2 = fmaf(y, y, c2);
            }
          } else {
            for (int e = 0; e < ds; ++e) {  // (wave-uniform trip count and lane index)
              const int i = j * ds + e;
              const float y = cb[i * 256 + (int)c];
              const int x0 = __builtin_amdgcn_readlane(__float_as_int(xv[0]), i & 63);
              const int x1 = __builtin_amdgcn_readlane(__float_as_int(xv[1]), i & 63);
              const float xx = __int_as_float(i < 64 ? x0 : x1);
              dot = fmaf(xx, y, dot);
              c2 = fmaf(y, y, c2);
            }
          }
          // (fused_lut4's arithmetic, operation for operation)
          float val = 2.f * dot;
          val = val - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q2v), j));
          val = val - c2;
          v += euclid ? val : dot;
        }
      }
      ex.insert_unsorted(want ? make_key(v + 0.0f, idx) : pad_key());
    }
    write_final<RM>(a, q, ex);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the next query overwrites the queue
  }
}

// pool mode, phase 2: the query's n_lists sorted lists of exact candidates (64 RX entries each, pads last) are
// merged BY RANK in LDS (rank_merge: fixed-step binary searches, eight lists in flight per lane) and the best k
// written out.  A flagged query (a pool or a list overflowed) is left to the exact kernel.
constexpr int kPoolMergeThreads = 512;
template <int NW>
__global__ __launch_bounds__(kPoolMergeThreads) void scan_pool_merge_kernel(ScanArgs a) {
  constexpr int RX = NW == 4 ? 8 : 4, LEN = 64 * RX;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int q = blockIdx.x;
  if (a.flags[q] == a.epoch) return;
  const int L = a.n_split * NW;
  unsigned* mhi = reinterpret_cast<unsigned*>(smem);
  unsigned* mlo = mhi + L * LEN;
  unsigned* ohi = mlo + L * LEN;
  const int kcap = (a.k + 63) / 64 * 64;
  unsigned* olo = ohi + kcap;
  for (int l = threadIdx.x >> 6; l < L; l += kPoolMergeThreads / 64) {
    const int64_t o = ((int64_t)q * L + l) * a.pool_cap;
    for (int e = threadIdx.x & 63; e < LEN; e += 64) {
      mhi[l * LEN + e] = a.pool_hi[o + e];
      mlo[l * LEN + e] = a.pool_lo[o + e];
    }
  }
  const Key pad = pad_key();
  for (int i = threadIdx.x; i < kcap; i += kPoolMergeThreads) {
    ohi[i] = pad.hi;
    olo[i] = pad.lo;
  }
  __syncthreads();
  rank_merge<LEN>(mhi, mlo, L, ohi, olo, kcap, (int)threadIdx.x, kPoolMergeThreads);
  __syncthreads();
  for (int e = threadIdx.x; e < a.k; e += kPoolMergeThreads) {
    const Key kk{ohi[e], olo[e]};
    const int idx = key_index(kk);
    const bool p = idx == kPadIdx;
    const int64_t o = (int64_t)q * a.k + e;
    a.out_vals[o] = p ? -INFINITY : key_value(kk);
    a.out_addr[o] = p ? -1 : (int64_t)idx;
    if (a.out_ids) a.out_ids[o] = p ? -1 : a.address2id[idx];
  }
}
// splits per query the ranking kernel's LDS (64 KiB) can take
static int pool_max_split(int m, int k) {
  const int nw = packed_waves(m), len = 64 * (nw == 4 ? 8 : 4);
  const int kcap = (k + 63) / 64 * 64;
  int s = (int)((65536 - (size_t)kcap * 8) / ((size_t)nw * len * 8));
  return s < 1 ? 1 : s;
}

// ---- host side -----------------------------------------------------------------------------

static int pow2_ceil(int r) {
  int p = 1;
  while (p < r) p <<= 1;
  return p;
}
static int list_regs(int k) { return pow2_ceil((k + 63) / 64); }  // 1, 2, 4, 8, 16
constexpr int kBandSlack = 8;  // spare list entries the packed path wants beyond k
static int list_regs_packed(int k) { return pow2_ceil((k + kBandSlack + 63) / 64); }
// ... and of the finish kernel's exact list on the dump routes: the band of the 16-bit table is ~70 table units wide
// whatever the values, and what lies within it below the k-th best grows with the slots scanned (k = 500 over 31 000
// slots: 504-520 survivors -- beyond 512 the query is redone by the exact kernel, 0.47 ms per 10 000 queries)
static int dump_finish_regs(int k, int64_t slots_hint) {
  // (up to 16 384 slots per query the extras stay within the 8 entries every packed path allows: k = 504 over 7 800
  // slots ran 2.58 ms against the lists' 3.36)
  const int slack = (slots_hint > 0 && slots_hint <= 16384) ? kBandSlack : 16 + k / 8;
  return pow2_ceil((k + slack + 63) / 64);
}
// Registers of the per-wave lists of the packed scan.  Tiles are dealt round-robin, so a wave's share
// of the top-k is ~k/NW: the lists are sized for at least 2k entries over the workgroup (64 RL per
// wave) instead of k + 8 per wave.  Folding 64 candidates into a 512- or 1024-entry sorted list used
// to dominate large k (k = 1000: 12.8 ms against 3.1 ms at k = 100, C2).  A wave that fills its list
// with candidates that still matter flags the query for the exact kernel (scan_packed_kernel).
#ifndef TPQ_SCAN_MIN_RL_K
#define TPQ_SCAN_MIN_RL_K 1  // experiment knob: below this k the lists keep the full k + 8
#endif
// The 2k budget counts on a cell's tiles being dealt to ALL the waves of the workgroup: the nearest cell alone
// can hold half of the top-k.  A cell much shorter than one round of tiles (waves x slots per tile: 512 slots at
// m = 64, 1024 at m = 32) lands in few waves -- on the reference's own benchmark grid (IVF4096 over 1 M vectors:
// 244 slots, ONE 256-slot tile at m <= 32) the 2k budget sent 1-2 % of the queries (93 % at n_probe = 1) through
// the exact redo at k = 100 (profiles/r04_reference_grid.json, "queries_redone_exactly") -- and gets 4k; so does
// a caller that gives no hint.  (Full-size lists everywhere would cost the long cells 10 % at k = 100, m <= 32.)
static int list_regs_scan(int k, int m, int max_nprobe, int64_t slots_hint, int waves = 0) {
  const int nw = waves ? waves : packed_waves(m);
  const int rp = list_regs_packed(k);
  if (k < TPQ_SCAN_MIN_RL_K) return rp;
  const int64_t round_slots = (int64_t)nw * 64 * packed_slots(m);
  // ("spread": the mean probed cell fills at least three quarters of a round of tiles)
  const bool spread = slots_hint > 0 && 4 * slots_hint >= 3 * round_slots * (max_nprobe > 0 ? max_nprobe : 1);
  const int budget = (spread ? 2 : 4) * k;
  int rl = 1;
  while (rl < rp && nw * 64 * rl < budget) rl <<= 1;
  return rl;
}

// pool mode (k > 248): registers of the threshold list (the wave's ceil(k / NW) best) and entries per pool
static int pool_list_regs(int k, int m) {
  const int nw = packed_waves(m);
  return pow2_ceil(((k + nw - 1) / nw + 63) / 64);
}
static int pool_capacity(int k, int m) {  // (16 / 32 registers per lane at read-back; four waves share a query's admissions)
  return (k <= 512 && m > 32) ? 1024 : 2048;
}  // (16 / 32 registers per lane at read-back)
static size_t pool_ws_bytes(int nq, int k, int m, int n_lists);

static size_t scan_lds_bytes_ref(int m, int R, int max_nprobe, int fused_floats) {
  const int lut_bytes = m * 1024;
  const int list_bytes = kScanWaves * R * 64 * 8;
  const int region0 = lut_bytes > list_bytes ? lut_bytes : list_bytes;
  size_t b = (size_t)region0 + kScanWaves * 512 + (size_t)(3 * max_nprobe + 1) * 4 + 4 +
             (size_t)fused_floats * 4;
  return (b + 15) & ~(size_t)15;
}
static size_t scan_lds_bytes_packed(int m, int R, int max_nprobe, int fused_floats, bool res) {
  const int nw = packed_waves(m);
  size_t b = (size_t)m * 1024 + packed_aux_bytes(R, m) + nw * 512 +
             (size_t)(3 * max_nprobe + 1) * 4 + 8 + 3 * nw * 4 + (res ? 8 * (size_t)max_nprobe : 0) +
             (size_t)fused_floats * 4;
  return (b + 15) & ~(size_t)15;
}
// dump modes: no un-permute rows; the 16-bit table is half the size and runs four waves per workgroup
static size_t scan_lds_bytes_dump(int m, bool sel16, int nw, int max_nprobe, int fused_floats) {
  size_t b = (size_t)m * (sel16 ? 512 : 1024) + nw * 512 + (size_t)(3 * max_nprobe + 1) * 4 + 8 + 3 * nw * 4 +
             (size_t)fused_floats * 4;
  return (b + 15) & ~(size_t)15;
}
static int fused_floats_of(const ScanArgs& a) { return a.lut ? 0 : a.m * a.ds + a.m; }
// fused finish (scan_packed_kernel RM > 0): instantiated for merged lists of up to kFuseMaxR registers
// (k <= 248); its merge buffers -- waves x 64 RM keys -- lie over the LUT, the un-permute rows and the queues
constexpr int kFuseMaxR = 4;
static bool fuse_fits(int m, int RM) {
  const int nw = packed_waves(m);
  return RM <= kFuseMaxR &&
         (size_t)(nw + 1) * RM * 64 * 8 <= (size_t)m * 1024 + packed_aux_bytes(RM, m) + (size_t)nw * 512;
}

// per-M translation units (scan_packed.hip compiled with -DTPQ_PACKED_M=<M>)
// (keep the list in sync with build.sh and torchpq_amd/kernels PACKED_M)
#define TPQ_PACKED_M_LIST(X) \
  X(4) X(8) X(12) X(16) X(20) X(24) X(28) X(32) X(40) X(48) X(56) X(64) X(96) X(120) X(128)
#define TPQ_DECLARE_PACKED(M) \
  int dispatch_packed_##M(const ScanArgs& a, const ResidualArgs* ra, int RL, int R, hipStream_t st); \
  int dispatch_pool_##M(const ScanArgs& a, int RL, hipStream_t st);                                  \
  int dispatch_dump_##M(const ScanArgs& a, int RL, int R, int mode, hipStream_t st);                 \
  int dump_occupancy_##M(int mode);
TPQ_PACKED_M_LIST(TPQ_DECLARE_PACKED)
#undef TPQ_DECLARE_PACKED

template <class K>
static int set_lds(K kernel, size_t bytes, const char* name) {
  if (bytes > 160 * 1024) {
    set_error("%s: needs %zu bytes of LDS (> 160 KiB per CU on gfx950)", name, bytes);
    return TPQ_ERR_UNSUPPORTED;
  }
  return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                   name);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace: [flags nq*4][delta nq*4][lists nq*n_lists*64R*8]; n_lists = n_split (reference
// kernel, only when n_split > 1) or n_split * waves-per-workgroup (packed kernel, always)
static size_t ws_bytes_for(int nq, int R, int n_lists) {
  return 2 * align256((size_t)nq * 4) + (size_t)nq * n_lists * R * 64 * 8 + align256((size_t)nq * n_lists * 4);
}

// pool mode workspace: [flags][delta][pool hi nq*n_lists*cap][pool lo ...][counts nq*n_lists]
static size_t pool_ws_bytes(int nq, int k, int m, int n_lists) {
  return 2 * align256((size_t)nq * 4) + (size_t)nq * n_lists * pool_capacity(k, m) * 8 +
         align256((size_t)nq * n_lists * 4);
}
static void fill_ws_pool(ScanArgs& a, void* workspace, int n_lists) {
  char* p = reinterpret_cast<char*>(workspace);
  a.flags = reinterpret_cast<int*>(p);
  a.ws_delta = reinterpret_cast<float*>(p + align256((size_t)a.nq * 4));
  char* pools = p + 2 * align256((size_t)a.nq * 4);
  a.pool_cap = pool_capacity(a.k, a.m);
  const size_t n = (size_t)a.nq * n_lists * a.pool_cap;
  a.pool_hi = reinterpret_cast<unsigned*>(pools);
  a.pool_lo = reinterpret_cast<unsigned*>(pools + n * 4);
  a.pool_cnt = reinterpret_cast<int*>(pools + n * 8);
}

static void fill_ws(ScanArgs& a, void* workspace, int R, int n_lists) {
  char* p = reinterpret_cast<char*>(workspace);
  a.flags = reinterpret_cast<int*>(p);
  a.ws_delta = reinterpret_cast<float*>(p + align256((size_t)a.nq * 4));
  char* lists = p + 2 * align256((size_t)a.nq * 4);
  a.ws_vals = reinterpret_cast<float*>(lists);
  a.ws_idx = reinterpret_cast<int*>(lists + (size_t)a.nq * n_lists * R * 64 * 4);
  a.list_evict = reinterpret_cast<int*>(lists + (size_t)a.nq * n_lists * R * 64 * 8);  // (dump modes)
}

static int validate(const ScanArgs& a) {
  TPQ_REQUIRE(a.codes && (a.lut || (a.query && a.codebook)) && a.cell_start && a.cell_size &&
                  a.n_probe_list && a.out_vals && a.out_addr,
              "ivfpq_scan: null pointer argument");
  TPQ_REQUIRE(a.lut || a.ds >= 1, "ivfpq_scan: bad sub-vector length %d", a.ds);
  TPQ_REQUIRE(a.nq >= 0 && a.max_nprobe >= 1, "ivfpq_scan: bad nq/max_nprobe (%d, %d)", a.nq,
              a.max_nprobe);
  TPQ_REQUIRE(a.m >= 4 && a.m % 4 == 0, "ivfpq_scan: n_subvectors=%d must be a positive multiple of 4", a.m);
  TPQ_REQUIRE(a.k >= 1 && a.k <= 1024, "ivfpq_scan: k=%d out of range (0, 1024]", a.k);
  TPQ_REQUIRE(a.n_slots >= 0 && a.n_slots < 0x7fffffffLL, "ivfpq_scan: n_slots=%lld out of range",
              (long long)a.n_slots);
  TPQ_REQUIRE(a.n_split >= 1 && a.n_split <= 1024, "ivfpq_scan: n_split=%d out of range", a.n_split);
  TPQ_REQUIRE((a.out_ids == nullptr) || (a.address2id != nullptr),
              "ivfpq_scan: out_ids given without address2id");
  return TPQ_OK;
}

static int need_ws(const void* ws, size_t have, size_t need, const char* who) {
  if (need && (!ws || have < need)) {
    set_error("%s: workspace too small (%zu < %zu)", who, have, need);
    return TPQ_ERR_WORKSPACE;
  }
  return TPQ_OK;
}

}  // namespace tpq
