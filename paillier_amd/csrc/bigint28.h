// bigint28.h -- fixed-width big-integer Montgomery arithmetic in radix 2^28 for gfx950.
//
// Why radix 2^28 and not 2^32 / 2^64 limbs: measured on MI355X (tools/ubench/valu_rates.hip,
// profiles/r01_valu_rates.txt) v_mad_u64_u32 issues every ~4 cycles per SIMD -- exactly the
// cost of a v_addc_co_u32.  A full-radix CIOS needs one carry instruction per multiply
// (the 64-bit accumulate can overflow and v_mad_u64_u32 has no carry-in), i.e. 2 issue slots
// per 32x32 product.  With 28-bit limbs a product is < 2^56, so a 64-bit accumulator absorbs
// 255 products without any carry handling: ONE v_mad_u64_u32 per product and nothing else in
// the inner loop.  (32/28)^2 = 1.31x more products, 2x fewer instructions per product.
//
// Representation: a number of WT = K*WL limbs, limb j weighs 2^(28 j).  "Lazy" limbs may
// exceed 2^28 slightly (<= 2^28 + 1); "canonical" = every limb < 2^28 and value < N.
// Montgomery radix R = 2^(28 WT) with 28 WT >= bits(N) + 3, so R > 4N and every Montgomery
// product of operands < 2N is again < 2N: no conditional subtraction except at the exit.
//
// Register budget: gfx950 VALU instructions address 256 architectural VGPRs.  One lane can
// hold WL 64-bit column accumulators (2 WL regs) + WL operand limbs: WL <= 74 -> 2048-bit
// moduli with one lane per number.  Wider moduli are split over K in {2,4,8} ADJACENT lanes of a
// quad: lane k owns columns [k*WL, (k+1)*WL).  Per outer row the lanes exchange one 64-bit
// accumulator (the column that crosses the slice boundary) and the Montgomery quotient digit
// by DPP quad_perm moves -- no LDS, no shuffles through memory.
//
// The `a` operand of a product is streamed one limb per row from LDS (wave-uniform row index,
// one column per number); the modulus limbs are read from LDS too.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PA_HD __host__ __device__ __forceinline__
#define PA_D __device__ __forceinline__
#else
#define PA_HD inline __attribute__((always_inline))
#endif

namespace pa28 {

constexpr int LB = 28;
constexpr uint32_t LMASK = (1u << LB) - 1u;

// total number of 28-bit limbs for a modulus of `bits` bits with the R > 4N slack
constexpr int limbs_for_bits(int bits) { return (bits + 3 + LB - 1) / LB; }

#if defined(__HIPCC__)

// d = a*b + c.  Written as asm because the C expression lets LICM hoist zext(b) out of the
// row loop, which doubles the registers held by the multiplicand array (measured: spills).
PA_D uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
  return d;
#else
  return (uint64_t)a * b + c;  // host pass of hipcc only parses this; it is never executed
#endif
}
PA_D uint64_t mul64(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
  return d;
#else
  return (uint64_t)a * b;
#endif
}

// quad_perm DPP controls.  K lanes of a number are adjacent and never straddle a quad.
template <int K> struct Dpp;
template <> struct Dpp<1> { static constexpr int kFromNext = 0xE4, kFromPrev = 0xE4, kBcast0 = 0xE4; };
template <> struct Dpp<2> {
  static constexpr int kFromNext = 1 | (1 << 2) | (3 << 4) | (3 << 6);  // [1,1,3,3]
  static constexpr int kFromPrev = 0 | (0 << 2) | (2 << 4) | (2 << 6);  // [0,0,2,2]
  static constexpr int kBcast0 = 0 | (0 << 2) | (2 << 4) | (2 << 6);    // [0,0,2,2]
};
template <> struct Dpp<4> {
  static constexpr int kFromNext = 1 | (2 << 2) | (3 << 4) | (3 << 6);  // [1,2,3,3]
  static constexpr int kFromPrev = 0 | (0 << 2) | (1 << 4) | (2 << 6);  // [0,0,1,2]
  static constexpr int kBcast0 = 0;                                       // [0,0,0,0]
};
// K == 8: the slices of a number fill half a 16-lane DPP row; neighbours by row_shl:1 / row_shr:1 (what crosses into another
// number is masked by is_first / not_last, as for the smaller shapes); the broadcast of slice 0 takes two steps (bcast0 below).
template <> struct Dpp<8> {
  static constexpr int kFromNext = 0x101;   // row_shl:1
  static constexpr int kFromPrev = 0x111;   // row_shr:1
  static constexpr int kBcast0 = 0;         // [0,0,0,0] inside each quad; the upper quad then takes it from 4 lanes down
};
template <int CTRL> PA_D uint32_t dpp_mov(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
#else
  return v;
#endif
}
// m of slice 0 of the same number
template <int K> PA_D uint32_t bcast0(uint32_t m) {
  uint32_t t = dpp_mov<Dpp<K>::kBcast0>(m);
  if (K == 8) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t u = dpp_mov<0x114>(t);   // row_shr:4
    t = (threadIdx.x & 4u) ? u : t;
#endif
  }
  return t;
}
template <int CTRL> PA_D uint64_t dpp_mov64(uint64_t v) {
  uint32_t lo = dpp_mov<CTRL>((uint32_t)v), hi = dpp_mov<CTRL>((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// ---------------------------------------------------------------------------------------
// Montgomery product  x <- a * x * R^-1 (mod N), lazy in (< 2N), lazy out (< 2N).
//   a limb i  = a_col[i * a_stride]       (LDS; same address for the K lanes of a number)
//   n_seg[j]  = modulus limb k*WL + j      (LDS; per-lane segment base)
//   k         = this lane's slice index, not_first/not_last = all-ones masks or 0
// Inner work per row and lane: 2 WL v_mad_u64_u32.  Accumulator bound: a column receives at
// most 2 products (< 2^56.001) per row while it travels WT rows, plus one carry < 2^37:
// (2 WT + 1) * 2^56 must stay below 2^64 -> WT <= 127 between flushes; for WT > 127 the upper
// bits of every accumulator are pushed one column up every 48 rows (kFlushEvery).
// ---------------------------------------------------------------------------------------
template <int WL, int K>
PA_D void montmul(uint32_t (&x)[WL], const uint32_t* a_col, int a_stride, const uint32_t* n_seg,
                  uint32_t n0inv, uint32_t is_first, uint32_t not_last) {
  constexpr int WT = WL * K;
  constexpr bool kNeedFlush = (2 * WT + 1) > 255;
  constexpr int kFlushEvery = 48;
  uint64_t t[WL];
#pragma unroll
  for (int j = 0; j < WL; ++j) t[j] = 0;

#pragma unroll 1
  for (int i = 0; i < WT; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");  // keep the modulus loads inside the row loop (no hoisting -> no spills)
#endif
    const uint32_t ai = a_col[i * a_stride];
#pragma unroll
    for (int j = 0; j < WL; ++j) t[j] = mad64(ai, x[j], t[j]);
    uint32_t m = ((uint32_t)t[0] * n0inv) & LMASK;
    if (K > 1) m = bcast0<K>(m);
    const uint64_t y0 = mad64(m, n_seg[0], t[0]);
    uint64_t c = y0 >> LB;
    if (K > 1) c &= (uint64_t)is_first | ((uint64_t)is_first << 32);
    t[0] = mad64(m, n_seg[1], t[1]) + c;
#pragma unroll
    for (int j = 2; j < WL; ++j) t[j - 1] = mad64(m, n_seg[j], t[j]);
    if (K > 1) {
      uint64_t in = dpp_mov64<Dpp<K>::kFromNext>(y0);
      t[WL - 1] = in & ((uint64_t)not_last | ((uint64_t)not_last << 32));
    } else {
      t[WL - 1] = 0;
    }
    if (kNeedFlush && (i % kFlushEvery) == kFlushEvery - 1) {
      // push bits >= 2^28 of every accumulator one column up (descending j: no carry chain)
      // The top column of a non-last lane hands its upper bits to column 0 of the next lane.
      // The last lane (or K == 1) has no column above: its top accumulator keeps its upper
      // bits and is flushed next time, when it has travelled down (it skips one flush only).
      uint64_t top_c = 0;
      if (K > 1) {
        const uint64_t nl = (uint64_t)not_last | ((uint64_t)not_last << 32);
        top_c = (t[WL - 1] >> LB) & nl;
        t[WL - 1] &= ((uint64_t)LMASK | ~nl);
      }
#pragma unroll
      for (int j = WL - 2; j >= 0; --j) {
        t[j + 1] += t[j] >> LB;
        t[j] &= LMASK;
      }
      if (K > 1) {
        const uint64_t nf = ~((uint64_t)is_first | ((uint64_t)is_first << 32));
        t[0] += dpp_mov64<Dpp<K>::kFromPrev>(top_c) & nf;
      }
    }
  }

  // lazy normalisation: s_j = lo_j + mid_{j-1} + hi_{j-2};  x_j = (s_j & M) + (s_{j-1} >> 28)
  uint32_t mid_in = 0, hi_in1 = 0, hi_in2 = 0;  // mid_{-1}, hi_{-1}, hi_{-2} from the previous lane
  if (K > 1) {
    uint32_t mid_top = (uint32_t)(t[WL - 1] >> LB) & LMASK;
    uint32_t hi_top = (uint32_t)(t[WL - 1] >> (2 * LB));
    uint32_t hi_top2 = (uint32_t)(t[WL - 2] >> (2 * LB));
    mid_in = dpp_mov<Dpp<K>::kFromPrev>(mid_top) & ~is_first;
    hi_in1 = dpp_mov<Dpp<K>::kFromPrev>(hi_top) & ~is_first;
    hi_in2 = dpp_mov<Dpp<K>::kFromPrev>(hi_top2) & ~is_first;
  }
  uint32_t s[WL];
#pragma unroll
  for (int j = 0; j < WL; ++j) {
    uint32_t lo = (uint32_t)t[j] & LMASK;
    uint32_t mid = j >= 1 ? ((uint32_t)(t[j - 1] >> LB) & LMASK) : mid_in;
    uint32_t hi = j >= 2 ? (uint32_t)(t[j - 2] >> (2 * LB)) : (j == 1 ? hi_in1 : hi_in2);
    s[j] = lo + mid + hi;
  }
  uint32_t s_in = 0;
  if (K > 1) s_in = dpp_mov<Dpp<K>::kFromPrev>(s[WL - 1] >> LB) & ~is_first;
#pragma unroll
  for (int j = 0; j < WL; ++j) x[j] = (s[j] & LMASK) + (j >= 1 ? (s[j - 1] >> LB) : s_in);
}

#endif  // __HIPCC__

}  // namespace pa28
