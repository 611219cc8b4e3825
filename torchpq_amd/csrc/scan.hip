// IVF list scan + top-k for gfx950 (MI355X).
//
// Replaces ivfpq_topk / ivfpq_top1 (torchpq/kernels/cuda/ivfpq_topk.cu:822-971,
// ivfpq_top1.cu:385-455) and their launchers (torchpq/kernels/IVFPQTopkCuda.py:81-142).
//
// Structure:
//   * one workgroup of 8 waves per (query, split); the query's 256-entry-per-sub-quantizer
//     LUT (m KB fp32) is staged in LDS; each wave walks its share of the probed cells in
//     64-slot tiles, one slot per lane, codes streamed straight from HBM to VGPRs;
//   * value(slot) = sum_j LUT[j][code_j] in fp32, ascending j from 0.f -- bit-identical to
//     consume_data (ivfpq_topk.cu:662-679);
//   * per-wave register top-k (wave_topk.h) + a workgroup-shared admission threshold in LDS;
//     no barrier inside the scan loop; one tree merge across the 8 waves at the end;
//   * n_split > 1 splits a query's tiles over several workgroups (small batches must still
//     fill 256 CUs; the reference's grid=(nq,) cannot) and a tiny merge kernel joins them.
//
// Two code layouts:
//   scan_ref_kernel    streams CellContainer._storage as is ([m/4][n_slots][4]); LDS lookups
//                      hit random banks (~3.5 cycles per half-wave access).
//   scan_packed_kernel streams the MI355X scan layout (pack.hip): per-slot XOR-permuted
//                      sub-quantizer order so the 32 lanes of a half-wave always read 32
//                      distinct banks (1 cycle).  The permuted order changes the fp32
//                      summation order, so it is used only as a conservative FILTER; every
//                      survivor (~1-2 % of slots) is re-evaluated in ascending-j order from
//                      the reference layout before it is ranked => results stay bit-identical.
#include "common.h"
#include "scan_layout.h"
#include "wave_topk.h"

namespace tpq {

constexpr int kScanWaves = 8;
constexpr int kScanThreads = kScanWaves * 64;

struct ScanArgs {
  const uint8_t* codes;    // reference layout [m/4][n_slots][4]
  const uint8_t* packed;   // scan layout (packed kernel only)
  const float* lut;        // [m][nq][256]
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
  int64_t n_slots;
  int nq, max_nprobe, m, k, n_split;
};

// ---- shared pieces -----------------------------------------------------------------------

struct ProbeTable {  // lives in LDS
  int* start;        // [max_nprobe]
  int* size;         // [max_nprobe]
  int* tile_begin;   // [max_nprobe + 1] exclusive prefix of ceil(size/64)
};

// wave 0 fills the probe table; cells whose start equals the previous probe's start are
// skipped (ivfpq_topk.cu:864-866)
__device__ __forceinline__ void build_probe_table(const ScanArgs& a, int q, int n_probe,
                                                  ProbeTable t) {
  const int lane = lane_id();
  int running = 0;
  for (int base = 0; base < n_probe; base += 64) {
    const int p = base + lane;
    int st = 0, sz = 0;
    if (p < n_probe) {
      st = (int)a.cell_start[(int64_t)q * a.max_nprobe + p];
      sz = (int)a.cell_size[(int64_t)q * a.max_nprobe + p];
      if (p > 0 && a.cell_start[(int64_t)q * a.max_nprobe + p - 1] == (int64_t)st) sz = 0;
      if (sz < 0) sz = 0;
    }
    int tiles = (sz + 63) >> 6;
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

template <int R>
__device__ __forceinline__ void store_list(const WaveTopK<R>& top, float* lv, int* li) {
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    lv[r * 64 + lane] = top.v[r];
    li[r * 64 + lane] = top.i[r];
  }
}

template <int R>
__device__ __forceinline__ void merge_list(WaveTopK<R>& top, const float* lv, const int* li) {
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < R; ++r) top.insert_sorted(lv[r * 64 + lane], li[r * 64 + lane]);
}

template <int R>
__device__ __forceinline__ void write_final(const ScanArgs& a, int q, const WaveTopK<R>& top) {
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * 64 + lane;
    if (e < a.k) {
      const int idx = top.i[r];
      const bool pad = (idx == kPadIdx);
      const int64_t adr = pad ? -1 : (int64_t)idx;
      a.out_vals[(int64_t)q * a.k + e] = pad ? -INFINITY : top.v[r];
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

__device__ __forceinline__ void stage_lut_linear(const ScanArgs& a, int q, float* lut) {
  // lut[j*256 + c] <- a.lut[(j*nq + q)*256 + c]; 16-byte loads, 1 KiB rows
  const float4* __restrict__ src = reinterpret_cast<const float4*>(a.lut);
  float4* dst = reinterpret_cast<float4*>(lut);
  for (int i = threadIdx.x; i < a.m * 64; i += kScanThreads) {
    const int j = i >> 6, c4 = i & 63;
    dst[i] = src[((int64_t)j * a.nq + q) * 64 + c4];
  }
}

// exact value of one slot in the reference's order (ascending j), LUT linear in LDS
__device__ __forceinline__ float exact_value_linear(const uint32_t* __restrict__ codes32,
                                                    int64_t n_slots, int s, int G,
                                                    const float* lut) {
  float v = 0.f;
  for (int g = 0; g < G; ++g) {
    const uint32_t w = codes32[(int64_t)g * n_slots + s];
    const float* row = lut + g * 1024;
    v += row[w & 255u];
    v += row[256 + ((w >> 8) & 255u)];
    v += row[512 + ((w >> 16) & 255u)];
    v += row[768 + (w >> 24)];
  }
  return v;
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

  const int q = blockIdx.x / a.n_split;
  const int part = blockIdx.x - q * a.n_split;
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  int n_probe = (int)a.n_probe_list[q];
  n_probe = n_probe < 0 ? 0 : (n_probe > a.max_nprobe ? a.max_nprobe : n_probe);

  if (wave == 0) {
    build_probe_table(a, q, n_probe, tab);
    if (lane == 0) *tau_key = f2key(-INFINITY);
  }
  stage_lut_linear(a, q, lut);
  __syncthreads();

  WaveSelector<R> sel;
  sel.init(qv_all + wave * 64, qi_all + wave * 64, a.k);
  NoRefine refine;

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
    const float tau_s = key2f(*reinterpret_cast<volatile unsigned*>(tau_key));
    sel.tau = fmaxf(sel.tau, tau_s);
    const float tau_before = sel.tau;
    sel.push(live && (v >= sel.tau), v, s, refine);
    if (sel.tau > tau_before && lane == 0) atomicMax(tau_key, f2key(sel.tau));
  }
  {
    const float tau_before = sel.tau;
    sel.flush(refine);
    if (sel.tau > tau_before && lane == 0) atomicMax(tau_key, f2key(sel.tau));
  }
  finish_query<R>(a, q, part, sel.top, reinterpret_cast<float*>(smem),
                  reinterpret_cast<int*>(smem + kScanWaves * R * 64 * 4));
}

// ---- packed-layout kernel ------------------------------------------------------------------
// LUT in LDS in block order: sub-quantizer block (base b, size B in {64,32,16,8,4}) occupies
// bytes [b*1024, (b+B)*1024): entry (j, c) at b*1024 + (c*B + (j-b))*4.  A slot at address s
// stores, at position p of block b, the code of sub-quantizer j = b + ((p-b) ^ (s & (B-1))),
// so lane (slot s) step p reads dword c*B + ((p-b) ^ (s&(B-1))): its bank differs from every
// other lane of the half-wave (consecutive s).

struct RefineExact {
  const uint32_t* codes32;
  int64_t n_slots;
  int G;
  const float* lut;  // packed-order LUT in LDS
  int m;
  __device__ __forceinline__ float operator()(float /*v*/, int idx, bool active) const {
    if (!active) return -INFINITY;
    float v = 0.f;
    int j = 0;
    for (int g = 0; g < G; ++g) {
      const uint32_t w = codes32[(int64_t)g * n_slots + idx];
#pragma unroll
      for (int u = 0; u < 4; ++u, ++j) {
        const unsigned c = (w >> (8 * u)) & 255u;
        v += lut[scan_layout::lut_dword(m, j, c)];
      }
    }
    return v;
  }
};

__device__ __forceinline__ void stage_lut_blocked(const ScanArgs& a, int q, float* lut) {
  // thread handles (j, 4 consecutive c): 16-byte global load, 4 scalar LDS stores.
  // Consecutive threads take consecutive j for the same c-group so that the LDS stores of a
  // half-wave land in distinct banks.
  const float4* __restrict__ src = reinterpret_cast<const float4*>(a.lut);
  const int m = a.m;
  for (int i = threadIdx.x; i < m * 64; i += kScanThreads) {
    const int c4 = i / m, j = i - c4 * m;
    const float4 x = src[((int64_t)j * a.nq + q) * 64 + c4];
    const int c = c4 * 4;
    lut[scan_layout::lut_dword(m, j, c + 0)] = x.x;
    lut[scan_layout::lut_dword(m, j, c + 1)] = x.y;
    lut[scan_layout::lut_dword(m, j, c + 2)] = x.z;
    lut[scan_layout::lut_dword(m, j, c + 3)] = x.w;
  }
}

template <int R, int M>
__global__ __launch_bounds__(kScanThreads) void scan_packed_kernel(ScanArgs a, float margin_rel) {
  using L = scan_layout::Layout<M>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int lut_bytes = M * 1024;
  constexpr int list_bytes = kScanWaves * R * 64 * 8;
  constexpr int region0 = lut_bytes > list_bytes ? lut_bytes : list_bytes;
  float* lut = reinterpret_cast<float*>(smem);
  float* qv_all = reinterpret_cast<float*>(smem + region0);
  int* qi_all = reinterpret_cast<int*>(smem + region0 + kScanWaves * 256);
  int* ptab = reinterpret_cast<int*>(smem + region0 + kScanWaves * 512);
  ProbeTable tab{ptab, ptab + a.max_nprobe, ptab + 2 * a.max_nprobe};
  unsigned* tau_key = reinterpret_cast<unsigned*>(ptab + 3 * a.max_nprobe + 1);
  float* red = reinterpret_cast<float*>(tau_key + 1);  // [kScanWaves] abs-max reduction

  const int q = blockIdx.x / a.n_split;
  const int part = blockIdx.x - q * a.n_split;
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  int n_probe = (int)a.n_probe_list[q];
  n_probe = n_probe < 0 ? 0 : (n_probe > a.max_nprobe ? a.max_nprobe : n_probe);

  if (wave == 0) {
    build_probe_table(a, q, n_probe, tab);
    if (lane == 0) *tau_key = f2key(-INFINITY);
  }
  stage_lut_blocked(a, q, lut);
  __syncthreads();

  // Error bound of the permuted-order fp32 sum against the ascending-order one:
  // |fast - exact| <= 2 (M-1) eps * sum_j max_c |LUT[j][c]|  (eps = 2^-24).  The filter admits
  // everything within that margin of the threshold; survivors are re-evaluated exactly.
  float amax = 0.f;
  for (int i = threadIdx.x; i < M * 256; i += kScanThreads) amax = fmaxf(amax, fabsf(lut[i]));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d, 64));
  if (lane == 0) red[wave] = amax;
  __syncthreads();
  amax = red[0];
#pragma unroll
  for (int w = 1; w < kScanWaves; ++w) amax = fmaxf(amax, red[w]);
  const float margin = margin_rel * (float)M * amax;

  WaveSelector<R> sel;
  sel.init(qv_all + wave * 64, qi_all + wave * 64, a.k);
  RefineExact refine{reinterpret_cast<const uint32_t*>(a.codes), a.n_slots, M / 4, lut, M};

  const int total_tiles = tab.tile_begin[n_probe];
  const int t_begin = (int)(((int64_t)total_tiles * part) / a.n_split);
  const int t_end = (int)(((int64_t)total_tiles * (part + 1)) / a.n_split);

  typename L::chunk_t w[L::kChunks];
  int p = 0;
  for (int T = t_begin + wave; T < t_end; T += kScanWaves) {
    while (T >= tab.tile_begin[p + 1]) ++p;
    const int off = ((T - tab.tile_begin[p]) << 6) + lane;
    const bool valid = off < tab.size[p];
    const int s = tab.start[p] + off;
    float v = 0.f;
    bool live = valid;
    if (valid) {
      if (a.is_empty) live = (a.is_empty[s] == 0);
      L::load(a.packed, a.n_slots, s, w);
      v = L::accumulate(w, s, lut);
    }
    const float tau_s = key2f(*reinterpret_cast<volatile unsigned*>(tau_key));
    sel.tau = fmaxf(sel.tau, tau_s);
    const float tau_before = sel.tau;
    sel.push(live && (v >= sel.tau - margin), v, s, refine);
    if (sel.tau > tau_before && lane == 0) atomicMax(tau_key, f2key(sel.tau));
  }
  {
    const float tau_before = sel.tau;
    sel.flush(refine);
    if (sel.tau > tau_before && lane == 0) atomicMax(tau_key, f2key(sel.tau));
  }
  finish_query<R>(a, q, part, sel.top, reinterpret_cast<float*>(smem),
                  reinterpret_cast<int*>(smem + kScanWaves * R * 64 * 4));
}

// ---- split merge ---------------------------------------------------------------------------

template <int R>
__global__ __launch_bounds__(64) void scan_merge_kernel(ScanArgs a) {
  const int q = blockIdx.x;
  WaveTopK<R> top;
  top.init();
  for (int part = 0; part < a.n_split; ++part) {
    const int64_t o = ((int64_t)q * a.n_split + part) * (R * 64);
    merge_list<R>(top, a.ws_vals + o, a.ws_idx + o);
  }
  write_final<R>(a, q, top);
}

// ---- host side -----------------------------------------------------------------------------

static int list_regs(int k) {
  int r = (k + 63) / 64;
  int p = 1;
  while (p < r) p <<= 1;
  return p;  // 1, 2, 4, 8, 16
}

static size_t scan_lds_bytes(int m, int R, int max_nprobe) {
  const int lut_bytes = m * 1024;
  const int list_bytes = kScanWaves * R * 64 * 8;
  const int region0 = lut_bytes > list_bytes ? lut_bytes : list_bytes;
  size_t b = (size_t)region0 + kScanWaves * 512 + (size_t)(3 * max_nprobe + 1) * 4 + 4 +
             kScanWaves * 4;
  return (b + 15) & ~(size_t)15;
}

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

static int validate(const ScanArgs& a, const void* ws, size_t ws_bytes) {
  TPQ_REQUIRE(a.codes && a.lut && a.cell_start && a.cell_size && a.n_probe_list && a.out_vals &&
                  a.out_addr,
              "ivfpq_scan: null pointer argument");
  TPQ_REQUIRE(a.nq >= 0 && a.max_nprobe >= 1, "ivfpq_scan: bad nq/max_nprobe (%d, %d)", a.nq,
              a.max_nprobe);
  TPQ_REQUIRE(a.m >= 4 && a.m % 4 == 0, "ivfpq_scan: n_subvectors=%d must be a positive multiple of 4", a.m);
  TPQ_REQUIRE(a.k >= 1 && a.k <= 1024, "ivfpq_scan: k=%d out of range (0, 1024]", a.k);
  TPQ_REQUIRE(a.n_slots >= 0 && a.n_slots < 0x7fffffffLL, "ivfpq_scan: n_slots=%lld out of range",
              (long long)a.n_slots);
  TPQ_REQUIRE(a.n_split >= 1 && a.n_split <= 1024, "ivfpq_scan: n_split=%d out of range", a.n_split);
  TPQ_REQUIRE((a.out_ids == nullptr) || (a.address2id != nullptr),
              "ivfpq_scan: out_ids given without address2id");
  if (a.n_split > 1) {
    const size_t need = tpq_ivfpq_scan_workspace_bytes(a.nq, a.k, a.n_split);
    if (!ws || ws_bytes < need) {
      set_error("ivfpq_scan: workspace too small (%zu < %zu)", ws_bytes, need);
      return TPQ_ERR_WORKSPACE;
    }
  }
  return TPQ_OK;
}

template <int R>
static int launch_ref(ScanArgs a, hipStream_t st) {
  const size_t lds = scan_lds_bytes(a.m, R, a.max_nprobe);
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

template <int R, int M>
static int launch_packed(ScanArgs a, hipStream_t st) {
  const size_t lds = scan_lds_bytes(M, R, a.max_nprobe);
  int rc = set_lds(scan_packed_kernel<R, M>, lds, "scan_packed_kernel");
  if (rc) return rc;
  const float margin_rel = 2.0f * 5.9604645e-8f * (float)(M - 1);
  hipLaunchKernelGGL((scan_packed_kernel<R, M>), dim3((unsigned)a.nq * a.n_split),
                     dim3(kScanThreads), lds, st, a, margin_rel);
  TPQ_LAUNCH_CHECK("scan_packed_kernel");
  if (a.n_split > 1) {
    hipLaunchKernelGGL(scan_merge_kernel<R>, dim3(a.nq), dim3(64), 0, st, a);
    TPQ_LAUNCH_CHECK("scan_merge_kernel");
  }
  return TPQ_OK;
}

template <int M>
static int dispatch_packed(const ScanArgs& a, int R, hipStream_t st) {
  switch (R) {
    case 1: return launch_packed<1, M>(a, st);
    case 2: return launch_packed<2, M>(a, st);
    case 4: return launch_packed<4, M>(a, st);
    case 8: return launch_packed<8, M>(a, st);
    default: return launch_packed<16, M>(a, st);
  }
}

}  // namespace tpq

using namespace tpq;

extern "C" size_t tpq_ivfpq_scan_workspace_bytes(int nq, int k, int n_split) {
  if (n_split <= 1 || nq <= 0 || k <= 0) return 0;
  return (size_t)nq * n_split * list_regs(k) * 64 * 8;
}

static void fill_ws(ScanArgs& a, void* workspace) {
  const int R = list_regs(a.k);
  a.ws_vals = reinterpret_cast<float*>(workspace);
  a.ws_idx = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) +
                                    (size_t)a.nq * a.n_split * R * 64 * 4);
}

extern "C" int tpq_ivfpq_scan_topk(const uint8_t* codes, const float* lut, const uint8_t* is_empty,
                                   const int64_t* cell_start, const int64_t* cell_size,
                                   const int64_t* n_probe_list, float* out_vals, int64_t* out_addr,
                                   const int64_t* address2id, int64_t* out_ids, int64_t n_slots,
                                   int nq, int max_nprobe, int m, int k, int n_split,
                                   void* workspace, size_t workspace_bytes, tpq_stream_t stream) {
  ScanArgs a{codes, nullptr, lut, is_empty, cell_start, cell_size, n_probe_list, out_vals, out_addr,
             address2id, out_ids, nullptr, nullptr, n_slots, nq, max_nprobe, m, k, n_split};
  int rc = validate(a, workspace, workspace_bytes);
  if (rc) return rc;
  if (nq == 0) return TPQ_OK;
  fill_ws(a, workspace);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (list_regs(k)) {
    case 1: return launch_ref<1>(a, st);
    case 2: return launch_ref<2>(a, st);
    case 4: return launch_ref<4>(a, st);
    case 8: return launch_ref<8>(a, st);
    default: return launch_ref<16>(a, st);
  }
}

extern "C" int tpq_ivfpq_scan_topk_packed(const uint8_t* packed, const uint8_t* codes,
                                          const float* lut, const uint8_t* is_empty,
                                          const int64_t* cell_start, const int64_t* cell_size,
                                          const int64_t* n_probe_list, float* out_vals,
                                          int64_t* out_addr, const int64_t* address2id,
                                          int64_t* out_ids, int64_t n_slots, int nq, int max_nprobe,
                                          int m, int k, int n_split, void* workspace,
                                          size_t workspace_bytes, tpq_stream_t stream) {
  ScanArgs a{codes, packed, lut, is_empty, cell_start, cell_size, n_probe_list, out_vals, out_addr,
             address2id, out_ids, nullptr, nullptr, n_slots, nq, max_nprobe, m, k, n_split};
  int rc = validate(a, workspace, workspace_bytes);
  if (rc) return rc;
  TPQ_REQUIRE(packed != nullptr, "ivfpq_scan_packed: null packed pointer");
  if (nq == 0) return TPQ_OK;
  fill_ws(a, workspace);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int R = list_regs(k);
  switch (m) {
    case 8: return dispatch_packed<8>(a, R, st);
    case 16: return dispatch_packed<16>(a, R, st);
    case 32: return dispatch_packed<32>(a, R, st);
    case 64: return dispatch_packed<64>(a, R, st);
    case 120: return dispatch_packed<120>(a, R, st);
    default:
      set_error("ivfpq_scan_packed: no packed kernel instantiated for n_subvectors=%d "
                "(available: 8, 16, 32, 64, 120); use tpq_ivfpq_scan_topk", m);
      return TPQ_ERR_UNSUPPORTED;
  }
}
