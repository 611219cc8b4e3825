// The MI355X "scan layout" of PQ codes and of the per-query LUT in LDS.
//
// Problem: the list scan does m LDS look-ups per slot with a data-dependent (random) code
// byte as index.  With the natural LUT[j][256] layout all 32 lanes of a half-wave look up the
// same sub-quantizer j at once, so their banks (= c % 32) are random: ~3.5 LDS cycles per
// ds_read_b32 instead of 1 (MI355X_MICROARCH: ds_read_b32 = 2 x 32 lanes, bank = dword % 32).
//
// Fix: make the lanes of a half-wave look up 32 DIFFERENT sub-quantizers at the same step and
// lay the LUT out as [code][sub-quantizer] so the bank is the sub-quantizer index:
//   * sub-quantizers are grouped in power-of-two blocks (greedy 64,...,64,32,16,8,4);
//     block (base b, size B) stores entry (j, c) at dword  b*256 + c*B + (j-b);
//   * the slot with address s keeps, at byte position p (b <= p < b+B), the code of
//     sub-quantizer  j = b + ((p-b) XOR (s mod B));
//   * lanes hold consecutive slots, so at step p lane s reads dword c*B + ((p-b)^(s%B)):
//     bank = ((p-b) ^ s) % 32 for B >= 32 -- 32 distinct banks for 32 consecutive s.
//     (B = 16/8/4 tails of an m that is not a multiple of 32 get bank = (c%(32/B))*B + ...,
//     i.e. at most a 32/B-way conflict on a fraction of the steps.)
// The bytes per slot are unchanged (m), only their order; packed shape [m/W][n_slots][W] with
// W = 16/8/4 so that one lane loads W contiguous bytes and a wave 64*W contiguous bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace tpq {
namespace scan_layout {

__host__ __device__ constexpr int chunk_width(int m) { return (m % 16 == 0) ? 16 : ((m % 8 == 0) ? 8 : 4); }

struct Block {
  int base, size;
};

// block containing sub-quantizer / position x (0 <= x < m)
__host__ __device__ constexpr Block block_of(int m, int x) {
  const int n64 = m / 64;
  if (x < n64 * 64) return Block{(x / 64) * 64, 64};
  int b = n64 * 64;
  const int rem = m - b;
  for (int B = 32; B >= 4; B >>= 1) {
    if (rem & B) {
      if (x < b + B) return Block{b, B};
      b += B;
    }
  }
  return Block{b, 4};
}

// LDS dword of LUT entry (sub-quantizer j, code c)
__host__ __device__ constexpr int lut_dword(int m, int j, int c) {
  const Block k = block_of(m, j);
  return k.base * 256 + c * k.size + (j - k.base);
}

// 16-bit selection table (scan_device.h "sel16"): HALFWORD of entry (j, c).  A 64-block keeps the two halves of its
// sub-quantizers interleaved -- halfword c*64 + 2*(j' & 31) + (j' >> 5), j' = j - base -- so that the dword, hence the
// bank, is c*32 + (j' & 31): the 32 lanes of a half-wave (32 distinct j' & 31) read 32 distinct banks.  Smaller blocks
// (tails of an m that is not a multiple of 64) are laid out plainly.
__host__ __device__ constexpr int lut16_halfword(int m, int j, int c) {
  const Block k = block_of(m, j);
  const int r = j - k.base;
  if (k.size == 64) return k.base * 256 + c * 64 + ((r & 31) << 1) + (r >> 5);
  return k.base * 256 + c * k.size + r;
}

// sub-quantizer stored at byte position p of the slot with address s (an involution in p)
__host__ __device__ constexpr int subq_at(int m, int p, int64_t s) {
  const Block k = block_of(m, p);
  return k.base + ((p - k.base) ^ (int)(s & (k.size - 1)));
}

// byte offset of position p of slot s inside the packed array
__host__ __device__ constexpr int64_t packed_offset(int m, int64_t n_slots, int p, int64_t s) {
  const int W = chunk_width(m);
  return ((int64_t)(p / W) * n_slots + s) * W + (p % W);
}

// compile-time block lookup usable inside fully unrolled loops
template <int M>
struct BlockAt {
  int base, size;
  __host__ __device__ constexpr explicit BlockAt(int p)
      : base(block_of(M, p).base), size(block_of(M, p).size) {}
};

// f(integral_constant<int, P>) for P = B ... E - 1: a loop whose index is a constant expression in the body
template <int B, int E, class F>
__device__ __forceinline__ void static_for_p(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_p<B + 1, E>(f);
  }
}

template <int W>
struct ChunkT;
template <>
struct ChunkT<16> {
  using type = uint4;
};
template <>
struct ChunkT<8> {
  using type = uint2;
};
template <>
struct ChunkT<4> {
  using type = uint32_t;
};

template <int M>
struct Layout {
  static constexpr int kW = chunk_width(M);
  static constexpr int kChunks = M / kW;
  using chunk_t = typename ChunkT<kW>::type;

  __device__ static __forceinline__ void load(const uint8_t* __restrict__ packed, int64_t n_slots,
                                              int s, chunk_t (&w)[kChunks]) {
    const chunk_t* __restrict__ src = reinterpret_cast<const chunk_t*>(packed);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) w[c] = src[(int64_t)c * n_slots + s];
  }
  // the tile loop's form: the slot index is UNSIGNED (no sign extension per address) and the caller clamps it with one
  // v_min_u32 against the last slot of the array (was: compare + select 0 -- the scan is VALU-issue-bound)
  __device__ static __forceinline__ void load_u(const uint8_t* __restrict__ packed, int64_t n_slots,
                                                uint32_t s, chunk_t (&w)[kChunks]) {
    const chunk_t* __restrict__ src = reinterpret_cast<const chunk_t*>(packed);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) w[c] = src[(int64_t)c * n_slots + (int64_t)(uint64_t)s];
  }

  __device__ static __forceinline__ uint32_t word(const chunk_t (&w)[kChunks], int d) {
    // d = dword index inside the slot (compile-time after unrolling)
    if constexpr (kW == 16) {
      const uint4& x = w[d >> 2];
      return (d & 3) == 0 ? x.x : (d & 3) == 1 ? x.y : (d & 3) == 2 ? x.z : x.w;
    } else if constexpr (kW == 8) {
      const uint2& x = w[d >> 1];
      return (d & 1) == 0 ? x.x : x.y;
    } else {
      return w[d];
    }
  }

  // sum over byte positions p ascending of LUT(subq_at(p, s), byte_p)  -- permuted order.
  // 64-blocks (256 B per code row) build the LDS byte address with ONE v_perm_b32:
  //   {0, 0, code byte, (s&63)<<2}  ^  (p<<2)   ==  code*256 + ((p ^ (s&63)) * 4)
  // (2 VALU + 1 add per look-up, no per-position address registers kept live).
  __device__ static __forceinline__ float accumulate(const chunk_t (&w)[kChunks], int s,
                                                     const float* __restrict__ lut) {
    // The fast value is used for selection only, so its summation order is free: two
    // interleaved partial sums (even / odd positions) halve the add chain and let the
    // compiler use v_pk_add_f32.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc = {0.f, 0.f};
    // lane constants of a 64-block: for each 16-position quarter h the low byte of the LDS
    // address, ((h ^ s[5:4]) << 6) | (s[3:0] << 2); the per-position remainder (p & 15) << 2 is
    // then an inline constant (<= 60) and the XOR fuses into one v_xad_u32.
    uint32_t lane64[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
      lane64[h] = ((((uint32_t)h ^ ((uint32_t)(s >> 4) & 3u)) << 6) | (((uint32_t)s & 15u) << 2));
    const char* __restrict__ lut_bytes = reinterpret_cast<const char*>(lut);
    // Blocks of 32 / 16 / 8 / 4 (every m but the multiples of 64; m = 8, 16, 32 are nothing else): byte address
    //   block base + (c << (log2 B + 2)) + (((rel ^ (s & (B-1))) << 2)
    // = (lane constant ^ ((rel & 15) << 2)) + (c << (log2 B + 2)),  lane constant = table + block base +
    //   (((s & (B-1)) << 2) ^ ((rel >> 4) << 6))   [one per block and 16-position quarter, hoisted out of the tile loop]
    // -- the byte shifted straight out of the code dword (SDWA), the XOR and the add in ONE v_xad_u32: two VALU per
    // look-up like the 64-blocks.  (Round 6; written as c * B + lane part, hipcc emitted the shift, a v_bitop3 XOR and a
    // v_lshl_add per look-up, plus an AND for byte 0: 3.25 VALU per look-up at m = 8 / 16 / 32, the instruction-bound
    // codes.)  The table sits at a multiple of 128 bytes in LDS (the start of the dynamic allocation) and a block base
    // is a multiple of 1 024: both commute with the XOR of the position bits.
    typedef const __attribute__((address_space(3))) float* lds_f32_ptr;
    const uint32_t lbase = (uint32_t)(uintptr_t)(lds_f32_ptr)lut;
    uint32_t shift_of[4];  // log2 B + 2 for B = 32, 16, 8, 4 (SDWA takes the shift from a register)
#pragma unroll
    for (int i = 0; i < 4; ++i) shift_of[i] = (uint32_t)(7 - i);
    static_for_p<0, M / 2>([&](auto p_c) {
      constexpr int p = 2 * decltype(p_c)::value;
      float t[2];
      static_for_p<0, 2>([&](auto u_c) {
        constexpr int u = decltype(u_c)::value;
        constexpr int pp = p + u;
        constexpr BlockAt<M> kb(pp);
        const uint32_t wd = word(w, pp >> 2);
        if constexpr (kb.size == 64) {
          constexpr int rel = pp - kb.base;
          constexpr uint32_t sel = 0x0c0c0000u | ((4u + (uint32_t)(pp & 3)) << 8);
          const uint32_t a = __builtin_amdgcn_perm(wd, lane64[rel >> 4], sel) ^ ((uint32_t)(rel & 15) << 2);
          t[u] = *reinterpret_cast<const float*>(lut_bytes + kb.base * 1024 + a);
        } else {
          constexpr int rel = pp - kb.base;
          constexpr int LOGB = kb.size == 32 ? 5 : (kb.size == 16 ? 4 : (kb.size == 8 ? 3 : 2));
          const uint32_t lane_c = lbase + (uint32_t)(kb.base * 1024) +
                                  ((((uint32_t)s & (uint32_t)(kb.size - 1)) << 2) ^ (uint32_t)((rel >> 4) << 6));
          const uint32_t a = sdwa_shl_xad<pp & 3, (rel & 15) << 2>(lane_c, shift_of[5 - LOGB], wd);
          t[u] = *(lds_f32_ptr)(uintptr_t)a;
        }
      });
      const f32x2 tv = {t[0], t[1]};
      acc += tv;
    });
    return acc.x + acc.y;
  }

  // (lane ^ IMM) + (byte BYTE of wd << shift): v_lshlrev_b32_sdwa + v_xad_u32.  Written as asm: hipcc, seeing that the
  // two parts share no bits, rewrites the sum into bfe + lshl_or + xor (accumulate16's note).
  template <int BYTE, int IMM>
  __device__ static __forceinline__ uint32_t sdwa_shl_xad(uint32_t lane, uint32_t shift, uint32_t wd) {
    static_assert(IMM >= 0 && IMM <= 64, "an inline constant");
    uint32_t a;
    if constexpr (BYTE == 0)
      asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
          "v_xad_u32 %0, %1, %4, %0" : "=&v"(a) : "v"(lane), "v"(shift), "v"(wd), "n"(IMM));
    else if constexpr (BYTE == 1)
      asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
          "v_xad_u32 %0, %1, %4, %0" : "=&v"(a) : "v"(lane), "v"(shift), "v"(wd), "n"(IMM));
    else if constexpr (BYTE == 2)
      asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
          "v_xad_u32 %0, %1, %4, %0" : "=&v"(a) : "v"(lane), "v"(shift), "v"(wd), "n"(IMM));
    else
      asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
          "v_xad_u32 %0, %1, %4, %0" : "=&v"(a) : "v"(lane), "v"(shift), "v"(wd), "n"(IMM));
    return a;
  }

  // The same walk over the 16-bit selection table (lut16_halfword): an exact integer sum of m u16 entries.
  // 64-blocks: byte address = c*128 + (((rel & 31) ^ (s & 31)) << 2) + (((rel >> 5) ^ (s >> 5 & 1)) << 1); per
  // 16-position quarter h the lane part is a constant, the position's low four bits an inline XOR, the code byte's
  // << 7 one shift of a sub-dword operand: (lane ^ imm) + (c << 7) -- two VALU per look-up as in accumulate(), and
  // the adds pair up in v_add3_u32.
  __device__ static __forceinline__ uint32_t accumulate16(const chunk_t (&w)[kChunks], int s,
                                                          const uint16_t* __restrict__ lut16) {
    uint32_t acc[2] = {0u, 0u};
    uint32_t lane64[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
      lane64[h] = ((((((uint32_t)h & 1u) << 4) ^ ((uint32_t)s & 16u)) | ((uint32_t)s & 15u)) << 2) |
                  ((((uint32_t)h >> 1) ^ (((uint32_t)s >> 5) & 1u)) << 1);
    // the table's LDS byte address as an INTEGER folded into the lane constants (a multiple of 128: it commutes with
    // the XOR of the position bits); the look-up then reads straight from the computed address -- formed as pointer +
    // offset, every look-up carried one more v_add_u32 of the (zero) base
    typedef const __attribute__((address_space(3))) uint16_t* lds_u16_ptr;
    const uint32_t lbase = (uint32_t)(uintptr_t)(lds_u16_ptr)lut16;
#pragma unroll
    for (int h = 0; h < 4; ++h) lane64[h] += lbase;
    const uint32_t seven = 7u;
    static_for_p<0, M>([&](auto p_c) {
      constexpr int pp = decltype(p_c)::value;
      constexpr BlockAt<M> kb(pp);
      const uint32_t wd = word(w, pp >> 2);
      uint32_t t;
      if constexpr (kb.size == 64) {
        constexpr int rel = pp - kb.base;
        // address = (lane part ^ position bits) + (code byte << 7): the byte shifted straight out of the code dword
        // (SDWA), the XOR and the add in one v_xad_u32.  Written as asm: hipcc, seeing that the two parts share no
        // bits, rewrites the sum into bfe + lshl_or + xor (+ an add3 with 0) -- three to four VALU per look-up
        uint32_t a;
        if constexpr ((pp & 3) == 0)
          asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
              "v_xad_u32 %0, %1, %4, %0" : "=&v"(a) : "v"(lane64[rel >> 4]), "v"(seven), "v"(wd), "n"((rel & 15) << 2));
        else if constexpr ((pp & 3) == 1)
          asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
              "v_xad_u32 %0, %1, %4, %0" : "=&v"(a) : "v"(lane64[rel >> 4]), "v"(seven), "v"(wd), "n"((rel & 15) << 2));
        else if constexpr ((pp & 3) == 2)
          asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
              "v_xad_u32 %0, %1, %4, %0" : "=&v"(a) : "v"(lane64[rel >> 4]), "v"(seven), "v"(wd), "n"((rel & 15) << 2));
        else
          asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
              "v_xad_u32 %0, %1, %4, %0" : "=&v"(a) : "v"(lane64[rel >> 4]), "v"(seven), "v"(wd), "n"((rel & 15) << 2));
        t = *(lds_u16_ptr)(uintptr_t)(a + (uint32_t)(kb.base * 512));
      } else {
        const uint32_t c = (wd >> (8 * (pp & 3))) & 255u;
        const int lane_part = (pp - kb.base) ^ (s & (kb.size - 1));
        t = lut16[kb.base * 256 + (int)c * kb.size + lane_part];
      }
      acc[(pp >> 1) & 1] += t;
    });
    return acc[0] + acc[1];
  }
};

}  // namespace scan_layout
}  // namespace tpq
