// kernels.hip -- gfx950 kernels of the batched Paillier engine.
//
//  vm_kernel<WL,K>   the hot kernel: a tiny wave-uniform "big-integer VM".  Every number of the
//                    batch (one lane, or K adjacent lanes for wide moduli) keeps ONE value x in
//                    VGPRs and executes the same host-generated program of Montgomery
//                    operations (LOAD/STORE/SQR/MUL/...).  A shared-exponent modexp, a window
//                    table build, a Horner reduction of a wide input, a per-lane-exponent modexp
//                    are all just programs; control flow is uniform, so there is no divergence.
//                    >= 99 % of a Paillier op is spent in its SQR/MUL (bigint28.h montmul).
//  k_*               light element-wise helpers (format conversion, canonicalisation, the exact
//                    division of L(), CRT recombination, g^m closed form).  One lane per number,
//                    operands streamed from limb-major arrays in HBM (coalesced).
//
// Batch layout in HBM ("limb-major"): uint32_t a[WT][NB]; limb l of number g at a[l*NB + g],
// 28-bit limbs (bigint28.h).  NB is the batch size padded to a multiple of 256.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bigint28.h"
#include "kernels.h"

using namespace pa28;

// ------------------------------------------------------------------------------------------
// VM kernel
// ------------------------------------------------------------------------------------------
template <int WL, int K>
__global__ void __launch_bounds__(VM_BLOCK, (3 * WL + 24 <= 256) ? 2 : 1) vm_kernel(VmArgs args) {
  constexpr int WT = WL * K;
  constexpr int NPB = VM_BLOCK / K;  // numbers per block
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* s_n = lds;                    // WT modulus limbs
  uint32_t* s_a = lds + ((WT + 3) & ~3);  // a-operand columns: [WT][NPB]

  const int segi = blockIdx.x < args.seg0_blocks ? 0 : (blockIdx.x - args.seg0_blocks < args.seg1_blocks ? 1 : 2);
  const VmSeg sg = args.seg[segi];
  const uint32_t blk = blockIdx.x - (segi >= 1 ? args.seg0_blocks : 0u) - (segi == 2 ? args.seg1_blocks : 0u);
  const int tid = threadIdx.x;
  const int k = tid % K;
  const int gl = tid / K;
  const size_t nb = sg.nb;
  const size_t g = (size_t)blk * NPB + gl;  // number index inside the segment (always < nb: nb is padded)
  const uint32_t is_first = (k == 0) ? 0xFFFFFFFFu : 0u;
  const uint32_t not_last = (k == K - 1) ? 0u : 0xFFFFFFFFu;

  for (int j = tid; j < WT; j += VM_BLOCK) s_n[j] = sg.nmod[j];
  __syncthreads();

  uint32_t x[WL];
#pragma unroll
  for (int j = 0; j < WL; ++j) x[j] = 0;
  size_t goff = 0;  // operand number offset (OP_SETOFF), used by tree products

  const uint32_t* prog = sg.prog;
  for (uint32_t pc = 0;; pc += 2) {
    const uint32_t w0 = prog[pc];
    const uint32_t arg = prog[pc + 1];
    const uint32_t op = w0 & 0xFFu;
    if (op == VM_END) break;
    if (op == VM_SETOFF) { goff = arg; continue; }
    if (op == VM_LOAD) {
      const uint32_t* p = sg.mem + ((size_t)arg * WT + (size_t)k * WL) * nb + g + goff;
#pragma unroll
      for (int j = 0; j < WL; ++j) x[j] = p[(size_t)j * nb];
      continue;
    }
    if (op == VM_LOADC) {
      const uint32_t* p = sg.consts + (size_t)arg * WT + (size_t)k * WL;
#pragma unroll
      for (int j = 0; j < WL; ++j) x[j] = p[j];
      continue;
    }
    if (op == VM_STORE) {
      uint32_t* p = sg.mem + ((size_t)arg * WT + (size_t)k * WL) * nb + g + goff;
#pragma unroll
      for (int j = 0; j < WL; ++j) p[(size_t)j * nb] = x[j];
      continue;
    }
    if (op == VM_STORET) {  // number-major inside the slot: [number][WT limbs] (tables gathered per number, see VM_MULV7)
      uint32_t* p = sg.mem + (size_t)arg * WT * nb + (g + goff) * WT + (size_t)k * WL;
#pragma unroll
      for (int j = 0; j < WL; ++j) p[j] = x[j];
      continue;
    }
    if (op == VM_ADD) {  // lazy limb-wise add; the program must renormalise with a MULC before squaring
      const uint32_t* p = sg.mem + ((size_t)arg * WT + (size_t)k * WL) * nb + g + goff;
#pragma unroll
      for (int j = 0; j < WL; ++j) x[j] += p[(size_t)j * nb];
      continue;
    }
    // ---- multiply family: stage the `a` operand into this number's LDS column, then montmul
    uint32_t* col = s_a + (size_t)(k * WL) * NPB + gl;
    if (op == VM_SQR) {
#pragma unroll
      for (int j = 0; j < WL; ++j) col[j * NPB] = x[j];
    } else if (op == VM_MULC) {
      const uint32_t* p = sg.consts + (size_t)arg * WT + (size_t)k * WL;
#pragma unroll
      for (int j = 0; j < WL; ++j) col[j * NPB] = p[j];
    } else if (op == VM_MULCV) {
      // fixed-base comb: table entry (window arg, digit of this number's exponent) of a table shared by the batch
      const uint32_t elimb = sg.digits[(size_t)(arg / 7u) * nb + g];
      const uint32_t digit = (elimb >> (4u * (arg % 7u))) & 15u;
      const uint32_t* p = sg.consts + ((size_t)((w0 >> 8) & 0x3FFFFFu) + 16u * arg + digit) * WT + (size_t)k * WL;
#pragma unroll
      for (int j = 0; j < WL; ++j) col[j * NPB] = p[j];
    } else if (op == VM_MULCV7) {
      const uint32_t elimb = sg.digits[(size_t)(arg / 4u) * nb + g];
      const uint32_t digit = (elimb >> (7u * (arg % 4u))) & 127u;
      const uint32_t* p = sg.consts + ((size_t)((w0 >> 8) & 0x3FFFFFu) + 128u * arg + digit) * WT + (size_t)k * WL;
#pragma unroll
      for (int j = 0; j < WL; ++j) col[j * NPB] = p[j];
    } else {
      size_t slot = arg;
      if (op == VM_MULV) {
        // arg = index of a 4-bit window of this number's own exponent (7 windows per 28-bit limb);
        // aux = first table slot.  The table index differs per number: a gather, not a branch.
        const uint32_t elimb = sg.digits[(size_t)(arg / 7u) * nb + g];
        slot = (size_t)((w0 >> 8) & 0x3FFFFFu) + ((elimb >> (4u * (arg % 7u))) & 15u);
      } else if (op == VM_MULV5) {
        // 5-bit windows of the exponent repacked as 25-bit words (5 windows per word); 32-entry table
        const uint32_t eword = sg.digits[(size_t)(arg / 5u) * nb + g];
        slot = (size_t)((w0 >> 8) & 0x3FFFFFu) + ((eword >> (5u * (arg % 5u))) & 31u);
      }
      if (op == VM_MULV7) {
        // 7-bit windows, 4 per 28-bit limb; 128-entry table whose slots are number-major (written by VM_STORET)
        const uint32_t elimb = sg.digits[(size_t)(arg / 4u) * nb + g];
        slot = (size_t)((w0 >> 8) & 0x3FFFFFu) + ((elimb >> (7u * (arg % 4u))) & 127u);
        const uint32_t* p = sg.mem + slot * WT * nb + (g + goff) * WT + (size_t)k * WL;
#pragma unroll
        for (int j = 0; j < WL; ++j) col[j * NPB] = p[j];
      } else {
        const uint32_t* p = sg.mem + (slot * WT + (size_t)k * WL) * nb + g + goff;
#pragma unroll
        for (int j = 0; j < WL; ++j) col[j * NPB] = p[(size_t)j * nb];
      }
    }
    montmul<WL, K>(x, s_a + gl, NPB, s_n + k * WL, sg.n0inv, is_first, not_last);
  }
}

template <int WL, int K>
static hipError_t launch_vm_t(const VmArgs& a, uint32_t blocks, hipStream_t st) {
  constexpr int WT = WL * K;
  const size_t lds = (size_t)(((WT + 3) & ~3) + WT * (VM_BLOCK / K)) * 4;
  // the attribute is per device: set it before every launch (a cheap host-side call) rather than caching one flag per
  // template instantiation, which would leave a second GPU of the same process without it
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)vm_kernel<WL, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((vm_kernel<WL, K>), dim3(blocks), dim3(VM_BLOCK), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_vm(int wl, int k, const VmArgs& a, uint32_t blocks, hipStream_t st) {
  if (wl == 37 && k == 1) return launch_vm_t<37, 1>(a, blocks, st);
  if (wl == 55 && k == 1) return launch_vm_t<55, 1>(a, blocks, st);
  if (wl == 74 && k == 1) return launch_vm_t<74, 1>(a, blocks, st);
  if (wl == 55 && k == 2) return launch_vm_t<55, 2>(a, blocks, st);
  if (wl == 74 && k == 2) return launch_vm_t<74, 2>(a, blocks, st);
  if (wl == 55 && k == 4) return launch_vm_t<55, 4>(a, blocks, st);
  if (wl == 74 && k == 4) return launch_vm_t<74, 4>(a, blocks, st);
  if (wl == 37 && k == 2) return launch_vm_t<37, 2>(a, blocks, st);
  if (wl == 37 && k == 4) return launch_vm_t<37, 4>(a, blocks, st);
  if (wl == 10 && k == 4) return launch_vm_t<10, 4>(a, blocks, st);   // 37-limb primes in four lanes per number (plan::prime_lanes)
  if (wl == 42 && k == 8) return launch_vm_t<42, 8>(a, blocks, st);   // n^3 of 3072-bit keys (9 216 bits): eight lanes per number
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// Helper kernels (one lane per number; generic width)
// ------------------------------------------------------------------------------------------

// Helper kernels are links of CHAINS between dependent ladders: tens of microseconds each, but next to a long ladder of a side lane
// whose waves run at s_setprio 3 .. 1 (gen_vm_asm.py fair_share) a priority-0 wave gets only the issue slots the ladder leaves --
// k_canon 0.9 ms, a three-product VM program 3.4 ms between the prover's a^n | x^n and Alpha launches (r04 trace).  They take the top
// priority for their few instructions.
#define CHAIN_PRIORITY() __builtin_amdgcn_s_setprio(3)


// big-endian bytes, element-major with fixed stride  ->  28-bit limbs, limb-major.
// Elements >= count are written as zero (padding lanes).
__global__ void k_unpack_be(const uint8_t* __restrict__ in, size_t stride, size_t nbytes, size_t count,
                            uint32_t* __restrict__ out, int wt, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  const uint8_t* p = in + g * stride;
  if (g < count && ((nbytes | stride | (size_t)(uintptr_t)in) & 3) == 0) {
    // whole 32-bit words through a sliding 64-bit window: one word load per 32 bits instead of five byte loads per limb
    const uint32_t* pw = (const uint32_t*)p;
    const size_t nw = nbytes / 4;
    uint64_t win = 0;
    int have = 0;
    size_t k = 0;
    for (int l = 0; l < wt; ++l) {
      if (have < LB && k < nw) {
        win |= (uint64_t)__builtin_bswap32(pw[nw - 1 - k]) << have;
        have += 32;
        ++k;
      }
      out[(size_t)l * nb + g] = (uint32_t)win & LMASK;
      win >>= LB;
      have = have > LB ? have - LB : 0;
    }
    return;
  }
  for (int l = 0; l < wt; ++l) {
    uint32_t v = 0;
    if (g < count) {
      // limb l covers bits [28 l, 28 l + 28): bytes (28 l)/8 .. from the little end
      size_t bit = (size_t)l * LB;
      size_t byte0 = bit / 8;
      uint64_t acc = 0;
      for (int b = 0; b < 5; ++b) {
        size_t bi = byte0 + b;  // index from the least significant byte
        if (bi < nbytes) acc |= (uint64_t)p[nbytes - 1 - bi] << (8 * b);
      }
      v = (uint32_t)(acc >> (bit % 8)) & LMASK;
    }
    out[(size_t)l * nb + g] = v;
  }
}

// canonical 28-bit limbs (limb-major) -> big-endian bytes, fixed stride.  Bits above nbytes are dropped
// (the caller guarantees the value fits).
__global__ void k_pack_be(const uint32_t* __restrict__ in, int wt, size_t nb, size_t count,
                          uint8_t* __restrict__ out, size_t stride, size_t nbytes) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  uint8_t* p = out + g * stride;
  if (((nbytes | stride | (size_t)(uintptr_t)out) & 3) == 0) {
    // whole 32-bit words: word w (from the most significant end) holds bits [8 (nbytes - 4 - 4 w), +32) of the number
    uint32_t* pw = (uint32_t*)p;
    const size_t nw = nbytes / 4;
    for (size_t w = 0; w < nw; ++w) {
      size_t bit = 8 * (nbytes - 4 - 4 * w);
      int l = (int)(bit / LB), sh = (int)(bit % LB);
      uint64_t v = 0;
      if (l < wt) v = in[(size_t)l * nb + g];
      if (l + 1 < wt) v |= (uint64_t)in[(size_t)(l + 1) * nb + g] << LB;
      if (l + 2 < wt && sh + 32 > 2 * LB) v |= (uint64_t)in[(size_t)(l + 2) * nb + g] << (2 * LB);
      pw[w] = __builtin_bswap32((uint32_t)(v >> sh));
    }
    return;
  }
  for (size_t bi = 0; bi < nbytes; ++bi) {  // bi = index from the least significant byte
    size_t bit = bi * 8;
    int l = (int)(bit / LB);
    int sh = (int)(bit % LB);
    uint64_t v = 0;
    if (l < wt) v = in[(size_t)l * nb + g];
    if (l + 1 < wt) v |= (uint64_t)in[(size_t)(l + 1) * nb + g] << LB;
    p[nbytes - 1 - bi] = (uint8_t)(v >> sh);
  }
}

// Lazy limbs (any limb < 2^32, value < 3N) -> canonical residue in [0, N), in place.
// Sequential carry, then up to two conditional subtractions of N.
__global__ void k_canon(uint32_t* __restrict__ x, const uint32_t* __restrict__ nmod, int wt, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint64_t c = 0;
  for (int l = 0; l < wt; ++l) {
    c += x[(size_t)l * nb + g];
    x[(size_t)l * nb + g] = (uint32_t)c & LMASK;
    c >>= LB;
  }
  for (int rep = 0; rep < 2; ++rep) {
    // compare x with N from the top
    int ge = 1;
    for (int l = wt - 1; l >= 0; --l) {
      uint32_t a = x[(size_t)l * nb + g], b = nmod[l];
      if (a != b) { ge = a > b; break; }
    }
    if (!ge) break;
    int32_t br = 0;
    for (int l = 0; l < wt; ++l) {
      int32_t d = (int32_t)x[(size_t)l * nb + g] - (int32_t)nmod[l] - br;
      br = d < 0;
      x[(size_t)l * nb + g] = (uint32_t)(d + (br << LB)) & LMASK;
    }
  }
}

// out[wo] = (a[wa] * b_const[wb] + addend) mod 2^(28 wo), canonical in, canonical out.  b is wave-uniform.
// Used for g^m = 1 + m*n and for m_p + p*h.
__global__ void k_mul_const_add(const uint32_t* __restrict__ a, int wa, const uint32_t* __restrict__ bconst, int wb,
                                const uint32_t* __restrict__ addv, int wadd, uint32_t add_small,
                                uint32_t* __restrict__ out, int wo, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint64_t acc = add_small, carry_hi = 0;  // 128-bit column accumulator (acc + carry_hi * 2^64)
  for (int c = 0; c < wo; ++c) {
    int i0 = c - (wb - 1) > 0 ? c - (wb - 1) : 0;
    int i1 = c < wa - 1 ? c : wa - 1;
    for (int i = i0; i <= i1; ++i) {
      uint64_t p = (uint64_t)a[(size_t)i * nb + g] * bconst[c - i];
      acc += p;
      carry_hi += acc < p;
    }
    if (addv && c < wadd) {
      uint64_t v = addv[(size_t)c * nb + g];
      acc += v;
      carry_hi += acc < v;
    }
    out[(size_t)c * nb + g] = (uint32_t)acc & LMASK;
    acc = (acc >> LB) | (carry_hi << (64 - LB));
    carry_hi >>= LB;
  }
}

// Exact division  l = t / d  with  t = u - sub_small - subv  (all canonical limb arrays, d odd, uniform):
//   l = t * dinv  mod 2^(28 wl),   dinv = d^-1 mod 2^(28 wl)
// The quotient is exact iff t >= 0 and l * d == t; otherwise status |= flag.
// Used for Paillier's L(u) = (u - 1) / n  (u = 1 mod n for every unit c; a non-unit c fails the check and is
// re-run on the generic path) and for floor((u-1)/n) = ((u-1) - ((u-1) mod n)) / n on the generic path.
// tbuf: scratch limb-major array [wu][nb] that receives t.
__global__ void k_div_exact(const uint32_t* __restrict__ u, int wu, uint32_t sub_small,
                            const uint32_t* __restrict__ subv, int wsub, uint32_t* __restrict__ tbuf,
                            const uint32_t* __restrict__ dinv, const uint32_t* __restrict__ d, int wd,
                            uint32_t* __restrict__ l, int wl, size_t nb, size_t count, int32_t* __restrict__ status,
                            int32_t flag) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  int32_t br = 0;
  for (int i = 0; i < wu; ++i) {
    int32_t v = (int32_t)u[(size_t)i * nb + g] - br - (i == 0 ? (int32_t)sub_small : 0) -
                ((subv && i < wsub) ? (int32_t)subv[(size_t)i * nb + g] : 0);
    br = v < 0;
    tbuf[(size_t)i * nb + g] = (uint32_t)(v + (br << LB)) & LMASK;
  }
  int bad = br;  // t < 0
  uint64_t acc = 0, hi = 0;
  for (int c = 0; c < wl; ++c) {
    for (int i = 0; i <= c; ++i) {
      uint64_t p = (uint64_t)(i < wu ? tbuf[(size_t)i * nb + g] : 0u) * dinv[c - i];
      acc += p;
      hi += acc < p;
    }
    l[(size_t)c * nb + g] = (uint32_t)acc & LMASK;
    acc = (acc >> LB) | (hi << (64 - LB));
    hi >>= LB;
  }
  acc = 0; hi = 0;
  for (int c = 0; c < wu; ++c) {
    int i0 = c - (wd - 1) > 0 ? c - (wd - 1) : 0;
    int i1 = c < wl - 1 ? c : wl - 1;
    for (int i = i0; i <= i1; ++i) {
      uint64_t p = (uint64_t)l[(size_t)i * nb + g] * d[c - i];
      acc += p;
      hi += acc < p;
    }
    bad |= ((uint32_t)acc & LMASK) != tbuf[(size_t)c * nb + g];
    acc = (acc >> LB) | (hi << (64 - LB));
    hi >>= LB;
  }
  bad |= (acc != 0) | (hi != 0);
  if (bad && g < count) status[g] |= flag;
}

// Register-resident versions of the two helpers on the Decrypt-2048 path (fixed widths, fully unrolled: the operand
// lives in VGPRs instead of being re-read from memory for every limb product).  A column of <= 128 products of
// canonical 28-bit limbs (or 74 products with one lazy 29-bit operand) plus the incoming carry stays below 2^64, so
// one 64-bit accumulator per column is enough.
template <int WU, int WL, int WD, bool CHECK>
__global__ void __launch_bounds__(256) k_div_exact_t(const uint32_t* __restrict__ u, uint32_t sub_small,
                                                     const uint32_t* __restrict__ subv,
                                                     const uint32_t* __restrict__ dinv, const uint32_t* __restrict__ d,
                                                     uint32_t* __restrict__ l, size_t nb, size_t count,
                                                     int32_t* __restrict__ status, int32_t flag) {
  CHAIN_PRIORITY();
  static_assert(WL <= 128 && WD <= 128, "single 64-bit column accumulator: <= 128 products of canonical limbs");
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint32_t t[WU];
  int32_t br = 0;
#pragma unroll
  for (int i = 0; i < WU; ++i) {
    int32_t v = (int32_t)u[(size_t)i * nb + g] - br - (i == 0 ? (int32_t)sub_small : 0) -
                ((subv && i < WL) ? (int32_t)subv[(size_t)i * nb + g] : 0);
    br = v < 0;
    t[i] = (uint32_t)(v + (br << LB)) & LMASK;
  }
  int bad = br;
  uint32_t q[WL];
  uint64_t acc = 0;
#pragma unroll
  for (int c = 0; c < WL; ++c) {
#pragma unroll
    for (int i = 0; i <= c; ++i) acc += (uint64_t)t[i] * dinv[c - i];
    q[c] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
#pragma unroll
  for (int c = 0; c < WL; ++c) l[(size_t)c * nb + g] = q[c];
  if (CHECK) {
    acc = 0;
#pragma unroll
    for (int c = 0; c < WU; ++c) {
#pragma unroll
      for (int i = (c - (WD - 1) > 0 ? c - (WD - 1) : 0); i <= (c < WL - 1 ? c : WL - 1); ++i) acc += (uint64_t)q[i] * d[c - i];
      bad |= ((uint32_t)acc & LMASK) != t[c];
      acc >>= LB;
    }
    bad |= acc != 0;
    if (bad && g < count) status[g] |= flag;
  }
}

// Exact division without the check, quotient narrower than the dividend (the digit split of the three-digit form:
// (X - X0) / n with X of WU limbs, the quotient known to fit WL limbs): l = t * dinv mod 2^(28 WL) needs t[0 .. WL) only.
template <int WU, int WL, int WS>
__global__ void __launch_bounds__(256) k_div_exact_nc(const uint32_t* __restrict__ u, uint32_t sub_small,
                                                      const uint32_t* __restrict__ subv, const uint32_t* __restrict__ dinv,
                                                      uint32_t* __restrict__ l, size_t nb) {
  CHAIN_PRIORITY();
  static_assert(WL <= 255 && WL <= WU, "a column of <= 255 products of canonical limbs fits the 64-bit accumulator");
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint32_t t[WL];
  int32_t br = 0;
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    int32_t v = (int32_t)u[(size_t)i * nb + g] - br - (i == 0 ? (int32_t)sub_small : 0) -
                ((subv && i < WS) ? (int32_t)subv[(size_t)i * nb + g] : 0);
    br = v < 0;
    t[i] = (uint32_t)(v + (br << LB)) & LMASK;
  }
  uint64_t acc = 0;
#pragma unroll
  for (int c = 0; c < WL; ++c) {
#pragma unroll
    for (int i = 0; i <= c; ++i) acc += (uint64_t)t[i] * dinv[c - i];
    l[(size_t)c * nb + g] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
}

template <int WA, int WB, int WO>
__global__ void __launch_bounds__(256) k_mul_const_add_t(const uint32_t* __restrict__ a, const uint32_t* __restrict__ bconst,
                                                         const uint32_t* __restrict__ addv, int wadd, uint32_t add_small,
                                                         uint32_t* __restrict__ out, size_t nb) {
  CHAIN_PRIORITY();
  static_assert(WA <= 74 || WB <= 74, "single 64-bit column accumulator: a column has min(WA, WB) <= 74 products of a 29-bit by a 28-bit limb, below 2^64");
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint32_t x[WA];
#pragma unroll
  for (int i = 0; i < WA; ++i) x[i] = a[(size_t)i * nb + g];
  uint64_t acc = add_small;
#pragma unroll
  for (int c = 0; c < WO; ++c) {
#pragma unroll
    for (int i = (c - (WB - 1) > 0 ? c - (WB - 1) : 0); i <= (c < WA - 1 ? c : WA - 1); ++i) acc += (uint64_t)x[i] * bconst[c - i];
    if (addv && c < wadd) acc += addv[(size_t)c * nb + g];
    out[(size_t)c * nb + g] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
}

// Runtime-width version for the wide products: one wave per block, the per-number operand staged limb-major in LDS
// ([wa][64], conflict-free) next to the uniform constant (broadcast reads).  Two-word column accumulator as in
// k_mul_const_add, so any widths are safe.
__global__ void __launch_bounds__(64) k_mul_const_add_lds(const uint32_t* __restrict__ a, int wa, const uint32_t* __restrict__ bconst,
                                                         int wb, const uint32_t* __restrict__ addv, int wadd, uint32_t add_small,
                                                         uint32_t* __restrict__ out, int wo, size_t nb) {
  CHAIN_PRIORITY();
  extern __shared__ uint32_t mca_lds[];
  uint32_t* xs = mca_lds;
  uint32_t* bs = mca_lds + (size_t)wa * 64;
  const int lane = threadIdx.x;
  const size_t g = (size_t)blockIdx.x * 64 + lane;
  for (int i = lane; i < wb; i += 64) bs[i] = bconst[i];
  if (g < nb)
    for (int i = 0; i < wa; ++i) xs[i * 64 + lane] = a[(size_t)i * nb + g];
  __syncthreads();
  if (g >= nb) return;
  uint64_t acc = add_small, hi = 0;
  for (int c = 0; c < wo; ++c) {
    const int i0 = c - (wb - 1) > 0 ? c - (wb - 1) : 0;
    const int i1 = c < wa - 1 ? c : wa - 1;
#pragma unroll 4
    for (int i = i0; i <= i1; ++i) {
      uint64_t p = (uint64_t)xs[i * 64 + lane] * bs[c - i];
      acc += p;
      hi += acc < p;
    }
    if (addv && c < wadd) {
      uint64_t v = addv[(size_t)c * nb + g];
      acc += v;
      hi += acc < v;
    }
    out[(size_t)c * nb + g] = (uint32_t)acc & LMASK;
    acc = (acc >> LB) | (hi << (64 - LB));
    hi >>= LB;
  }
}

// Wide products of a per-number operand by a uniform constant as a LOOP over blocks of KB output columns (the fully unrolled
// templates above are 100 - 300 KB of straight-line code at these widths and run at the speed of the instruction fetch: 1.1 ms for
// 148 x 74 limbs on 16 384 numbers, 2.8 ms for the truncated 148 x 148).  One wave per block; the operand x sits limb-major in LDS
// ([wx][64], conflict-free), the constant comes through wave-uniform loads.  For a block of KB columns and KB consecutive limbs of x
// the 2 KB - 1 constant limbs it needs are fetched once: 3 KB - 1 loads for KB^2 multiplies.  One 64-bit accumulator per column:
// min(wx, wb) products of a <= 29-bit by a 28-bit limb must stay below 2^64 (min <= 110; 148 for canonical operands -- the launcher checks).
//   out[c] = sum_i x[i] b[c - i] (+ addv[c]) (+ add_small at c = 0), c < wo;   x = a - sub_small - subv (borrows propagated) when `sub`.
template <int KB>
__global__ void __launch_bounds__(64) k_mul_blocked(const uint32_t* __restrict__ a, int wx, bool sub, uint32_t sub_small,
                                                    const uint32_t* __restrict__ subv, int wsub, const uint32_t* __restrict__ bconst,
                                                    int wb, const uint32_t* __restrict__ addv, int wadd, uint32_t add_small,
                                                    uint32_t* __restrict__ out, int wo, size_t nb) {
  CHAIN_PRIORITY();
  extern __shared__ uint32_t mb_lds[];
  uint32_t* xs = mb_lds;                        // [wx][64]
  const int lane = threadIdx.x;
  const size_t g = (size_t)blockIdx.x * 64 + lane;
  if (g < nb) {
    if (sub) {
      int32_t br = 0;
      for (int i = 0; i < wx; ++i) {
        int32_t v = (int32_t)a[(size_t)i * nb + g] - br - (i == 0 ? (int32_t)sub_small : 0) -
                    ((subv && i < wsub) ? (int32_t)subv[(size_t)i * nb + g] : 0);
        br = v < 0;
        xs[i * 64 + lane] = (uint32_t)(v + (br << LB)) & LMASK;
      }
    } else {
      for (int i = 0; i < wx; ++i) xs[i * 64 + lane] = a[(size_t)i * nb + g];
    }
  } else {
    for (int i = 0; i < wx; ++i) xs[i * 64 + lane] = 0;
  }
  __syncthreads();
  if (g >= nb) return;
  uint64_t carry = add_small;
  for (int c0 = 0; c0 < wo; c0 += KB) {
    uint64_t acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = 0;
    const int i_lo = c0 - (wb - 1) > 0 ? c0 - (wb - 1) : 0;
    const int i_hi = c0 + KB - 1 < wx - 1 ? c0 + KB - 1 : wx - 1;
    for (int i0 = i_lo; i0 <= i_hi; i0 += KB) {
      uint32_t xi[KB], bw[2 * KB - 1];
#pragma unroll
      for (int ii = 0; ii < KB; ++ii) xi[ii] = i0 + ii <= i_hi ? xs[(i0 + ii) * 64 + lane] : 0u;
#pragma unroll
      for (int t = 0; t < 2 * KB - 1; ++t) {
        const int j = c0 - i0 + t - (KB - 1);           // = (c0 + k) - (i0 + ii) with t = k - ii + KB - 1
        bw[t] = (j >= 0 && j < wb) ? bconst[j] : 0u;
      }
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int ii = 0; ii < KB; ++ii) acc[k] += (uint64_t)xi[ii] * bw[k - ii + KB - 1];
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int c = c0 + k;
      if (c < wo) {
        uint64_t v = acc[k] + carry;
        if (addv && c < wadd) v += addv[(size_t)c * nb + g];
        out[(size_t)c * nb + g] = (uint32_t)v & LMASK;
        carry = v >> LB;
      }
    }
  }
}

// out = x - 1 (w limbs); x == 0 wraps to all-ones limbs (callers flag that lane separately)
__global__ void k_sub_one(const uint32_t* __restrict__ x, uint32_t* __restrict__ out, int w, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  int32_t br = 1;
  for (int l = 0; l < w; ++l) {
    int32_t v = (int32_t)x[(size_t)l * nb + g] - br;
    br = v < 0;
    out[(size_t)l * nb + g] = (uint32_t)(v + (br << LB)) & LMASK;
  }
}

// clear every bit >= `bits` of a w-limb number (r mod 2^bits)
__global__ void k_mask_bits(uint32_t* __restrict__ x, int w, size_t nb, size_t bits) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  for (int l = 0; l < w; ++l) {
    size_t lo = (size_t)l * LB;
    uint32_t m = lo >= bits ? 0u : (bits - lo >= (size_t)LB ? LMASK : ((1u << (bits - lo)) - 1u));
    x[(size_t)l * nb + g] &= m;
  }
}

// out[wa+wb] = a[wa] * b[wb] as plain integers (canonical limbs in, canonical limbs out); both operands per-number.
// Used only for the unreduced c^4 and c_i^2 that feed the Fiat-Shamir hash (thresholdkey.go:241,248) -- 16 384-bit and 8 192-bit
// integers.  Two steps: (1) one thread per (number, output column): the column's sum of limb products, flushed into (28-bit limb,
// carry) every 64 terms so that nothing overflows 64 bits; a square (a == b) takes every pair once, doubled.  (2) one thread per
// number walks the columns and resolves the carries.  (One thread per number for the whole product -- round 4 -- is 256 waves of
// 586 x 586 dependent loads: 9 ms for the c^4 of 16 384 ciphertexts; this form: 1.3 ms.)
__global__ void k_comb7_transpose(const uint32_t* __restrict__ mem, size_t nb, int wt, int nwin, uint32_t first, uint32_t* __restrict__ table) {
  // one thread per (limb, window, digit): reads are coalesced over the windows' lanes only for small nwin -- a one-off per key
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nwin * 128 * (size_t)wt;
  if (t >= total) return;
  const int l = (int)(t % (size_t)wt);
  const size_t e = t / (size_t)wt;                 // 128 i + d
  const size_t i = e >> 7, d = e & 127;
  table[((size_t)first + e) * (size_t)wt + l] = mem[((1 + d) * (size_t)wt + (size_t)l) * nb + i];
}

__global__ void __launch_bounds__(256) k_mul_plain_cols(const uint32_t* __restrict__ a, int wa, const uint32_t* __restrict__ b, int wb, int sq,
                                                        uint32_t* __restrict__ lo, uint64_t* __restrict__ cy, size_t nb) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)blockIdx.y;
  if (g >= nb) return;
  const int i0 = c - (wb - 1) > 0 ? c - (wb - 1) : 0;
  int i1 = c < wa - 1 ? c : wa - 1;
  uint64_t acc = 0, carry = 0;
  uint32_t low = 0;
  int n = 0;
  auto flush = [&] {
    const uint64_t t = acc + low;
    low = (uint32_t)t & LMASK;
    carry += t >> LB;
    acc = 0;
    n = 0;
  };
  if (sq) {
    const int ih = (c - 1) >> 1;                              // pairs (i, c - i) with i < c - i   (c = 0: none)
    if (c > 0 && ih < i1) i1 = ih;
    if (c > 0)
      for (int i = i0; i <= i1; ++i) {
        acc += 2ull * ((uint64_t)a[(size_t)i * nb + g] * a[(size_t)(c - i) * nb + g]);
        if (++n == 64) flush();
      }
    if (!(c & 1) && (c >> 1) < wa) {
      const uint64_t d = a[(size_t)(c >> 1) * nb + g];
      acc += d * d;
    }
  } else {
    for (int i = i0; i <= i1; ++i) {
      acc += (uint64_t)a[(size_t)i * nb + g] * b[(size_t)(c - i) * nb + g];
      if (++n == 128) flush();
    }
  }
  flush();
  lo[(size_t)c * nb + g] = low;
  cy[(size_t)c * nb + g] = carry;
}
__global__ void k_mul_plain_carry(const uint32_t* __restrict__ lo, const uint64_t* __restrict__ cy, int wo, uint32_t* __restrict__ out, size_t nb) {
  CHAIN_PRIORITY();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint64_t carry = 0;
  for (int c = 0; c < wo; ++c) {
    const uint64_t t = carry + lo[(size_t)c * nb + g];
    out[(size_t)c * nb + g] = (uint32_t)t & LMASK;
    carry = (t >> LB) + cy[(size_t)c * nb + g];
  }
}

// 32-byte big-endian digests (uint32 words, limb-major [8][nb]) -> 10 canonical 28-bit limbs (the integer E)
__global__ void k_digest_to_limbs(const uint32_t* __restrict__ dg, uint32_t* __restrict__ out, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  // value = sum_i word[i] * 2^(32 (7 - i))
  for (int l = 0; l < 10; ++l) {
    int bit = l * LB;
    int w = bit / 32, sh = bit % 32;               // little-endian word index
    uint64_t v = 0;
    if (w < 8) v = dg[(size_t)(7 - w) * nb + g];
    if (w + 1 < 8) v |= (uint64_t)dg[(size_t)(7 - (w + 1)) * nb + g] << 32;
    out[(size_t)l * nb + g] = (uint32_t)(v >> sh) & LMASK;
  }
}

// flags[g] = (x == 0)
__global__ void k_is_zero(const uint32_t* __restrict__ x, int w, size_t nb, int32_t* __restrict__ flags) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  uint32_t o = 0;
  for (int l = 0; l < w; ++l) o |= x[(size_t)l * nb + g];
  flags[g] = o == 0;
}

// status[g] |= flag where the w-limb number x is not exactly 1 (canonical limbs).  Used on the first digit of
// x^(p-1) in pair form: it is 1 for every unit and 0 or p for a multiple of p.
__global__ void k_flag_not_one(const uint32_t* __restrict__ x, int w, size_t nb, size_t count, int32_t* __restrict__ status,
                               int32_t flag) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  uint32_t o = x[g] ^ 1u;
  for (int l = 1; l < w; ++l) o |= x[(size_t)l * nb + g];
  if (o) status[g] |= flag;
}

// x <- c (uniform constant, w limbs) on the lanes whose flag is set
__global__ void k_select_const(const int32_t* __restrict__ flags, const uint32_t* __restrict__ c, uint32_t* __restrict__ x,
                               int w, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb || !flags[g]) return;
  for (int l = 0; l < w; ++l) x[(size_t)l * nb + g] = c[l];
}

// out = (a - b) mod q for canonical a, b in [0, q): a - b + (a < b ? q : 0)
__global__ void k_sub_mod(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                          const uint32_t* __restrict__ q, uint32_t* __restrict__ out, int w, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  int32_t br = 0;
  for (int l = 0; l < w; ++l) {
    int32_t d = (int32_t)a[(size_t)l * nb + g] - (int32_t)b[(size_t)l * nb + g] - br;
    br = d < 0;
    out[(size_t)l * nb + g] = (uint32_t)(d + (br << LB)) & LMASK;
  }
  if (br) {
    uint32_t c = 0;
    for (int l = 0; l < w; ++l) {
      uint32_t s = out[(size_t)l * nb + g] + q[l] + c;
      out[(size_t)l * nb + g] = s & LMASK;
      c = s >> LB;
    }
  }
}

// copy `w` limbs (zero-extending to wo) between limb-major arrays, optionally starting at source limb `l0`
__global__ void k_copy_limbs(const uint32_t* __restrict__ in, int l0, int w, uint32_t* __restrict__ out, int wo, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  for (int l = 0; l < wo; ++l) out[(size_t)l * nb + g] = (l < w) ? in[(size_t)(l0 + l) * nb + g] : 0u;
}

// the same for `nchunks` consecutive w-limb pieces of `in` at once: piece k -> out + k * out_stride (wo limbs each, zero-extended)
__global__ void k_copy_chunks(const uint32_t* __restrict__ in, int w, int nchunks, uint32_t* __restrict__ out, size_t out_stride,
                              int wo, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  for (int k = 0; k < nchunks; ++k)
    for (int l = 0; l < wo; ++l) out[(size_t)k * out_stride + (size_t)l * nb + g] = (l < w) ? in[(size_t)(k * w + l) * nb + g] : 0u;
}

// fill a limb-major array with a uniform constant (wo limbs)
__global__ void k_fill_const(const uint32_t* __restrict__ c, uint32_t* __restrict__ out, int wo, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  for (int l = 0; l < wo; ++l) out[(size_t)l * nb + g] = c[l];
}

// gather / scatter of selected numbers (used to re-run flagged lanes on the generic path)
__global__ void k_gather(const uint32_t* __restrict__ in, size_t nb_in, const uint32_t* __restrict__ idx, size_t n_idx,
                         uint32_t* __restrict__ out, size_t nb_out, int w) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb_out) return;
  for (int l = 0; l < w; ++l) out[(size_t)l * nb_out + g] = (g < n_idx) ? in[(size_t)l * nb_in + idx[g]] : 0u;
}
__global__ void k_scatter(const uint32_t* __restrict__ in, size_t nb_in, const uint32_t* __restrict__ idx, size_t n_idx,
                          uint32_t* __restrict__ out, size_t nb_out, int w) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_idx) return;
  for (int l = 0; l < w; ++l) out[(size_t)l * nb_out + idx[g]] = in[(size_t)l * nb_in + g];
}

// out[l][g] (stride nb_out) = g < count ? in[l][g] (stride nb_in) : fill[l]   (re-stride / pad a batch)
__global__ void k_restride(const uint32_t* __restrict__ in, size_t nb_in, size_t count, const uint32_t* __restrict__ fill,
                           uint32_t* __restrict__ out, size_t nb_out, int w) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb_out) return;
  for (int l = 0; l < w; ++l) out[(size_t)l * nb_out + g] = (g < count) ? in[(size_t)l * nb_in + g] : (fill ? fill[l] : 0u);
}

// out[g] = g < half ? lo[g] : hi[g - half]   for g < 2*half   (all arrays limb-major with stride nb)
__global__ void k_merge_halves(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, size_t half,
                               uint32_t* __restrict__ out, size_t nb, int w) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= 2 * half) return;
  const uint32_t* src = g < half ? lo + g : hi + (g - half);
  for (int l = 0; l < w; ++l) out[(size_t)l * nb + g] = src[(size_t)l * nb];
}

// ------------------------------------------------------------------------------------------
// SHA-256 of a transcript of big integers, one lane per transcript (random_oracle.go:20-32, thresholdkey.go:319-326):
// the message is the concatenation of gmp.Int.Bytes() of each part -- minimal big-endian, NO length prefix, zero
// contributes no bytes.  Parts are canonical 28-bit-limb numbers in limb-major arrays.
// ------------------------------------------------------------------------------------------
struct ShaPart { const uint32_t* p; int w; };
struct ShaArgs { ShaPart part[6]; int nparts; };

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

__constant__ uint32_t kSha256K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

struct Sha256 {
  uint32_t h[8];
  uint32_t w[16];
  uint32_t fill;      // bytes in the current block
  uint64_t total;     // message bytes
  __device__ void init() {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
    h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
    fill = 0; total = 0;
    for (int i = 0; i < 16; ++i) w[i] = 0;
  }
  __device__ void block() {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    uint32_t ws[16];
    for (int i = 0; i < 16; ++i) ws[i] = w[i];
    for (int i = 0; i < 64; ++i) {
      uint32_t wi;
      if (i < 16) wi = ws[i];
      else {
        uint32_t w15 = ws[(i + 1) & 15], w2 = ws[(i + 14) & 15];
        uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
        uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
        wi = ws[i & 15] + s0 + ws[(i + 9) & 15] + s1;
        ws[i & 15] = wi;
      }
      uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
      uint32_t ch = (e & f) ^ (~e & g);
      uint32_t t1 = hh + S1 + ch + kSha256K[i] + wi;
      uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
      uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    for (int i = 0; i < 16; ++i) w[i] = 0;
    fill = 0;
  }
  __device__ void put(uint8_t byte) {
    w[fill >> 2] |= (uint32_t)byte << (24 - 8 * (fill & 3));
    ++fill; ++total;
    if (fill == 64) block();
  }
  // a whole big-endian word at once; only while the block position is word-aligned
  __device__ void put_word(uint32_t x) {
    w[fill >> 2] = x;
    fill += 4; total += 4;
    if (fill == 64) block();
  }
  __device__ void finish() {
    uint64_t bits = total * 8;
    put(0x80); --total;
    if (fill > 56) { while (fill != 0) { put(0); --total; } }
    while (fill < 56) { put(0); --total; }
    w[14] = (uint32_t)(bits >> 32);
    w[15] = (uint32_t)bits;
    block();
  }
};

// digest_out: uint32[8][nb] limb-major (word i of number g at [i*nb + g]); bit_out (optional): low bit of the digest
// read as a big-endian integer (RandomOracleBit, random_oracle.go:10-16)
__global__ void k_sha256_transcript(ShaArgs a, size_t nb, size_t count, uint32_t* __restrict__ digest_out,
                                    int32_t* __restrict__ bit_out) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  Sha256 s;
  s.init();
  for (int pi = 0; pi < a.nparts; ++pi) {
    const uint32_t* p = a.part[pi].p;
    const int w = a.part[pi].w;
    // find the most significant non-zero byte
    long top = -1;
    for (int l = w - 1; l >= 0; --l) {
      uint32_t v = p[(size_t)l * nb + g];
      if (v) { top = (long)l * LB + (31 - __clz(v)); break; }
    }
    if (top < 0) continue;                       // zero: Bytes() is empty
    auto byte_at = [&](long byte) {
      long bit = byte * 8;
      int l = (int)(bit / LB), sh = (int)(bit % LB);
      uint64_t v = p[(size_t)l * nb + g];
      if (l + 1 < w) v |= (uint64_t)p[(size_t)(l + 1) * nb + g] << LB;
      return (uint8_t)(v >> sh);
    };
    long byte = top / 8;
    // bytes one at a time up to a word boundary of the block, then whole words (the transcript of a DDLEQ instance is 2 KB per
    // lane and sits between Alpha and the response on the prover's critical path: 1.8 ms byte by byte), then the tail
    while (byte >= 0 && (s.fill & 3)) s.put(byte_at(byte--));
    if (byte >= 3) {
      long bit = (byte - 3) * 8;                 // lowest bit of the word whose top byte is `byte`
      int l = (int)(bit / LB), sh = (int)(bit % LB);
      for (; byte >= 3; byte -= 4) {
        uint64_t v = p[(size_t)l * nb + g];
        if (l + 1 < w) v |= (uint64_t)p[(size_t)(l + 1) * nb + g] << LB;
        if (l + 2 < w) v |= (uint64_t)p[(size_t)(l + 2) * nb + g] << (2 * LB);
        s.put_word((uint32_t)(v >> sh));
        sh -= 32;
        while (sh < 0) { sh += LB; --l; }
      }
    }
    while (byte >= 0) s.put(byte_at(byte--));
  }
  s.finish();
  if (digest_out)
    for (int i = 0; i < 8; ++i) digest_out[(size_t)i * nb + g] = s.h[i];
  if (bit_out) bit_out[g] = (int32_t)(s.h[7] & 1u);
}

// ok[g] = (a == b) limb-wise (canonical numbers)
__global__ void k_equal(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, int w, size_t nb, size_t count,
                        int32_t* __restrict__ ok) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  uint32_t d = 0;
  for (int l = 0; l < w; ++l) d |= a[(size_t)l * nb + g] ^ b[(size_t)l * nb + g];
  ok[g] = d == 0;
}

// out <- flags[g] ? a : b   (w limbs)
__global__ void k_select(const int32_t* __restrict__ flags, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                         uint32_t* __restrict__ out, int w, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  const uint32_t* src = flags[g] ? a : b;
  for (int l = 0; l < w; ++l) out[(size_t)l * nb + g] = src[(size_t)l * nb + g];
}


// flags[g] = (gcd(x, N) != 1) for canonical x and the odd modulus N: a per-lane binary GCD on limb-major work arrays
// (work: [2][w][nb]).  SLOW PATH ONLY -- it runs when the one inversion of batch_inverse's product tree fails, i.e. when
// some element of the batch is not a unit (hostile or malformed input), to find out WHICH lanes those are; mpz_invert
// gives the reference the same information per call.  O(bits) iterations of O(w) limb operations per lane, divergent.
__global__ void k_unit_flags(const uint32_t* __restrict__ x, const uint32_t* __restrict__ nmod, int w, size_t nb, size_t count,
                             uint32_t* __restrict__ work, int32_t* __restrict__ flags) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  if (g >= count) { flags[g] = 0; return; }
  uint32_t* A = work + g;
  uint32_t* B = work + (size_t)w * nb + g;
  int la = 0, lb = 0;
  for (int l = 0; l < w; ++l) {
    const uint32_t v = x[(size_t)l * nb + g], m = nmod[l];
    A[(size_t)l * nb] = v;
    B[(size_t)l * nb] = m;
    if (v) la = l + 1;
    if (m) lb = l + 1;
  }
  // invariant: B is odd.  gcd(A, B) is preserved by A >>= 1 (A even) and by (A, B) <- (|A - B|, min(A, B)).
  while (la > 0) {
    const uint32_t a0 = A[0];
    if (a0 == 0) {                               // a whole zero limb: shift right by 28 bits
      for (int l = 0; l + 1 < la; ++l) A[(size_t)l * nb] = A[(size_t)(l + 1) * nb];
      A[(size_t)(la - 1) * nb] = 0;
      --la;
      continue;
    }
    const int s = __builtin_ctz(a0);
    if (s) {
      uint32_t cur = a0;
      for (int l = 0; l < la; ++l) {
        const uint32_t nxt = (l + 1 < la) ? A[(size_t)(l + 1) * nb] : 0u;
        A[(size_t)l * nb] = ((cur >> s) | (nxt << (LB - s))) & LMASK;
        cur = nxt;
      }
      if (A[(size_t)(la - 1) * nb] == 0) --la;
    }
    int cmp = la - lb;
    if (cmp == 0)
      for (int l = la - 1; l >= 0; --l) {
        const uint32_t av = A[(size_t)l * nb], bv = B[(size_t)l * nb];
        if (av != bv) { cmp = av > bv ? 1 : -1; break; }
      }
    if (cmp == 0) break;                         // A == B: the gcd is B
    if (cmp < 0) {
      uint32_t* t = A; A = B; B = t;
      const int tl = la; la = lb; lb = tl;
    }
    int32_t br = 0;                              // A -= B  (A > B, both odd: the difference is even)
    for (int l = 0; l < la; ++l) {
      const int32_t d = (int32_t)A[(size_t)l * nb] - (l < lb ? (int32_t)B[(size_t)l * nb] : 0) - br;
      br = d < 0;
      A[(size_t)l * nb] = (uint32_t)(d + (br << LB)) & LMASK;
    }
    while (la > 0 && A[(size_t)(la - 1) * nb] == 0) --la;
  }
  flags[g] = !(lb == 1 && B[0] == 1u);
}

// out[l][g] = table[idx[g]][l]: per-number exponent limbs from a small table of exponents (one row per key share)
__global__ void k_gather_rows(const uint32_t* __restrict__ table, int w, const int32_t* __restrict__ idx, size_t count,
                              uint32_t* __restrict__ out, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  const uint32_t* row = table + (size_t)(g < count ? idx[g] : 0) * w;
  for (int l = 0; l < w; ++l) out[(size_t)l * nb + g] = row[l];
}

// exponent limbs (28 bits) -> 25-bit words: word k = bits [25 k, 25 k + 25) of the number
__global__ void k_repack_windows5(const uint32_t* __restrict__ in, int we, uint32_t* __restrict__ out, int we5, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  for (int k = 0; k < we5; ++k) {
    const int bit = 25 * k, l = bit / LB, sh = bit % LB;
    uint64_t v = l < we ? in[(size_t)l * nb + g] : 0u;
    if (l + 1 < we) v |= (uint64_t)in[(size_t)(l + 1) * nb + g] << LB;
    out[(size_t)k * nb + g] = (uint32_t)(v >> sh) & ((1u << 25) - 1u);
  }
}

// status[g] |= flag where flags[g] != 0 (g < count)
__global__ void k_or_flags(const int32_t* __restrict__ flags, size_t count, int32_t* __restrict__ status, int32_t flag) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < count && flags[g]) status[g] |= flag;
}

// ok[g] = 0 where flags[g] != 0
__global__ void k_clear_where(const int32_t* __restrict__ flags, size_t count, int32_t* __restrict__ ok) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < count && flags[g]) ok[g] = 0;
}

// one wave per block: the helpers are latency-bound at the batch sizes of the level-two / threshold paths (16384 lanes =
// 256 one-wave blocks = every CU), and bandwidth-bound ones lose nothing
// ---- wire format (Ciphertext.Bytes() / NewCiphertextFromBytes, paillier.go:374-401: encoding/gob) -----------------------------
// HBM-bound byte movers; one 64-lane block per ciphertext, lanes stride over its bytes (coalesced within an element).
// unpack: the big-endian magnitude of element g is len[g] bytes at src + off[g] (found by the host's walk over the gob messages);
// it lands right-aligned in the element's fixed stride, zero-padded on the left.
__global__ void k_bytes_gather_be(const uint8_t* __restrict__ src, const uint64_t* __restrict__ off, const uint32_t* __restrict__ len,
                                  size_t count, uint8_t* __restrict__ out, size_t stride) {
  CHAIN_PRIORITY();
  const size_t g = blockIdx.x;
  if (g >= count) return;
  const uint8_t* s = src + off[g];
  const size_t n = len[g], pad = stride - n;
  uint8_t* d = out + g * stride;
  for (size_t i = threadIdx.x; i < stride; i += blockDim.x) d[i] = i < pad ? (uint8_t)0 : s[i - pad];
}
// pack, step 1: significant bytes of every fixed-stride big-endian element (0 for the value 0)
__global__ void k_be_lengths(const uint8_t* __restrict__ in, size_t stride, size_t count, uint32_t* __restrict__ len) {
  CHAIN_PRIORITY();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  const uint8_t* p = in + g * stride;
  size_t z = 0;
  while (z < stride && p[z] == 0) ++z;
  len[g] = (uint32_t)(stride - z);
}
// gob's unsigned integer: one byte below 128, else the negated byte count followed by the big-endian bytes
__device__ inline uint32_t gob_uint(uint8_t* d, uint32_t v) {
  if (v < 128) { d[0] = (uint8_t)v; return 1; }
  const uint32_t nb = v < (1u << 8) ? 1 : v < (1u << 16) ? 2 : v < (1u << 24) ? 3 : 4;
  d[0] = (uint8_t)(256 - nb);
  for (uint32_t i = 0; i < nb; ++i) d[1 + i] = (uint8_t)(v >> (8 * (nb - 1 - i)));
  return 1 + nb;
}
// pack, step 2: blob g at dst + off[g] = prefix (the two type-definition messages: constant) | length of the value message |
// head (type id, field delta of C) | length of GobEncode() | version byte | magnitude | tail (Level / EncMethod fields, end)
__global__ void k_gob_emit(const uint8_t* __restrict__ in, size_t stride, const uint32_t* __restrict__ len, const uint64_t* __restrict__ off,
                           size_t count, const uint8_t* __restrict__ prefix, uint32_t prefix_len, const uint8_t* __restrict__ head,
                           uint32_t head_len, const uint8_t* __restrict__ tail, uint32_t tail_len, uint8_t* __restrict__ dst) {
  CHAIN_PRIORITY();
  const size_t g = blockIdx.x;
  if (g >= count) return;
  __shared__ uint32_t s_at;
  uint8_t* d = dst + off[g];
  const uint32_t n = len[g], glen = n + 1;
  for (uint32_t i = threadIdx.x; i < prefix_len; i += blockDim.x) d[i] = prefix[i];
  if (threadIdx.x == 0) {
    uint8_t tmp[8];
    const uint32_t lg = gob_uint(tmp, glen);
    uint32_t at = prefix_len;
    at += gob_uint(d + at, head_len + lg + glen + tail_len);          // length of the value message
    for (uint32_t i = 0; i < head_len; ++i) d[at + i] = head[i];
    at += head_len;
    at += gob_uint(d + at, glen);
    d[at++] = 2;                                                       // GobEncode: version 1 << 1 | sign 0
    s_at = at;
  }
  __syncthreads();
  const uint32_t at = s_at;
  const uint8_t* p = in + g * stride + (stride - n);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) d[at + i] = p[i];
  for (uint32_t i = threadIdx.x; i < tail_len; i += blockDim.x) d[at + n + i] = tail[i];
}

#define HELPER_GRID(nb) dim3((unsigned)(((nb) + 63) / 64)), dim3(64)

void launch_unpack_be(const uint8_t* in, size_t stride, size_t nbytes, size_t count, uint32_t* out, int wt, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_unpack_be, HELPER_GRID(nb), 0, st, in, stride, nbytes, count, out, wt, nb);
}
void launch_pack_be(const uint32_t* in, int wt, size_t nb, size_t count, uint8_t* out, size_t stride, size_t nbytes, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_be, HELPER_GRID(count ? count : 1), 0, st, in, wt, nb, count, out, stride, nbytes);
}
void launch_canon(uint32_t* x, const uint32_t* nmod, int wt, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_canon, HELPER_GRID(nb), 0, st, x, nmod, wt, nb);
}
void launch_mul_const_add(const uint32_t* a, int wa, const uint32_t* bconst, int wb, const uint32_t* addv, int wadd,
                          uint32_t add_small, uint32_t* out, int wo, size_t nb, hipStream_t st) {
  if (wa == 37 && wb == 37 && wo == 74 && (!addv || wadd <= 74)) {
    hipLaunchKernelGGL((k_mul_const_add_t<37, 37, 74>), HELPER_GRID(nb), 0, st, a, bconst, addv, wadd, add_small, out, nb);
    return;
  }
  // (the Teichmueller lift's 1 + p (z a) + p^2 (...) and the exit from the digit form modulo p^3, 2048-bit keys)
  if (wa == 74 && wb == 37 && wo == 110 && (!addv || wadd <= 110)) {
    hipLaunchKernelGGL((k_mul_const_add_t<74, 37, 110>), HELPER_GRID(nb), 0, st, a, bconst, addv, wadd, add_small, out, nb);
    return;
  }
  if (wa == 37 && wb == 74 && wo == 110 && (!addv || wadd <= 110)) {
    hipLaunchKernelGGL((k_mul_const_add_t<37, 74, 110>), HELPER_GRID(nb), 0, st, a, bconst, addv, wadd, add_small, out, nb);
    return;
  }
  // wide products on the blocked loop kernel (see k_mul_blocked): the closed form of (1 + n)^k modulo n^3 (148 x 74), the exit from
  // the three-digit form (74 x 148), Garner modulo p^3 q^3 (110 x 110)
  if ((size_t)wa * wb >= 74u * 110u && (wa < wb ? wa : wb) <= 110 && (size_t)wa * 256 <= 60 * 1024 && (!addv || wadd <= wo)) {
    hipLaunchKernelGGL((k_mul_blocked<8>), dim3((unsigned)((nb + 63) / 64)), dim3(64), (size_t)wa * 256, st, a, wa, false, 0u,
                       (const uint32_t*)nullptr, 0, bconst, wb, addv, wadd, add_small, out, wo, nb);
    return;
  }
  // (the closed form of (1 + n)^k modulo n^3 of a 2048-bit key: 1 + k n and + C(k, 2) n^2)
  if (wa == 148 && wb == 74 && wo == 220 && (!addv || wadd <= 220)) {
    hipLaunchKernelGGL((k_mul_const_add_t<148, 74, 220>), HELPER_GRID(nb), 0, st, a, bconst, addv, wadd, add_small, out, nb);
    return;
  }
  if (wa == 74 && wb == 148 && wo == 220 && (!addv || wadd <= 220)) {
    hipLaunchKernelGGL((k_mul_const_add_t<74, 148, 220>), HELPER_GRID(nb), 0, st, a, bconst, addv, wadd, add_small, out, nb);
    return;
  }
  if (wa == 55 && wb == 55 && wo == 110 && (!addv || wadd <= 110)) {
    hipLaunchKernelGGL((k_mul_const_add_t<55, 55, 110>), HELPER_GRID(nb), 0, st, a, bconst, addv, wadd, add_small, out, nb);
    return;
  }
  if (wa == 74 && wb == 74 && wo == 148 && (!addv || wadd <= 148)) {
    hipLaunchKernelGGL((k_mul_const_add_t<74, 74, 148>), HELPER_GRID(nb), 0, st, a, bconst, addv, wadd, add_small, out, nb);
    return;
  }
  // wide products (the three-digit form's exit F2 n^2 + ..., the closed form of (1+n)^m mod n^3): too long to unroll
  // into registers, so the operand is staged in LDS and the block is one wave (a 16384-lane batch still covers 256 CUs)
  if ((size_t)wa * 64 * 4 + (size_t)wb * 4 <= 60 * 1024) {
    size_t lds = (size_t)wa * 64 * 4 + (size_t)wb * 4;
    hipLaunchKernelGGL(k_mul_const_add_lds, dim3((unsigned)((nb + 63) / 64)), dim3(64), lds, st, a, wa, bconst, wb, addv, wadd,
                       add_small, out, wo, nb);
    return;
  }
  hipLaunchKernelGGL(k_mul_const_add, HELPER_GRID(nb), 0, st, a, wa, bconst, wb, addv, wadd, add_small, out, wo, nb);
}
void launch_div_exact(const uint32_t* u, int wu, uint32_t sub_small, const uint32_t* subv, int wsub, uint32_t* tbuf,
                      const uint32_t* dinv, const uint32_t* d, int wd, uint32_t* l, int wl, size_t nb, size_t count,
                      int32_t* status, int32_t flag, hipStream_t st) {
  // status == nullptr: the caller knows the division is exact (digit split): no check
  if (wu == 74 && wl == 37 && wd == 37 && (!subv || wsub == 37)) {
    if (status)
      hipLaunchKernelGGL((k_div_exact_t<74, 37, 37, true>), HELPER_GRID(nb), 0, st, u, sub_small, subv, dinv, d, l, nb, count,
                         status, flag);
    else
      hipLaunchKernelGGL((k_div_exact_t<74, 37, 37, false>), HELPER_GRID(nb), 0, st, u, sub_small, subv, dinv, d, l, nb, count,
                         status, flag);
    return;
  }
  if (wu == 148 && wl == 74 && wd == 74 && (!subv || wsub == 74)) {
    if (status)
      hipLaunchKernelGGL((k_div_exact_t<148, 74, 74, true>), HELPER_GRID(nb), 0, st, u, sub_small, subv, dinv, d, l, nb, count,
                         status, flag);
    else
      hipLaunchKernelGGL((k_div_exact_t<148, 74, 74, false>), HELPER_GRID(nb), 0, st, u, sub_small, subv, dinv, d, l, nb, count,
                         status, flag);
    return;
  }
  if (wu == 110 && wl == 55 && wd == 55 && (!subv || wsub == 55)) {
    if (status)
      hipLaunchKernelGGL((k_div_exact_t<110, 55, 55, true>), HELPER_GRID(nb), 0, st, u, sub_small, subv, dinv, d, l, nb, count,
                         status, flag);
    else
      hipLaunchKernelGGL((k_div_exact_t<110, 55, 55, false>), HELPER_GRID(nb), 0, st, u, sub_small, subv, dinv, d, l, nb, count,
                         status, flag);
    return;
  }
  if (status && !subv && wu == 110 && wl == 74 && wd == 37) {
    // L_p = (u - 1) / p for u modulo p^3 of a 2048-bit key: the level-two CRT decryption and the Teichmueller lift of the DDLEQ
    // prover (the generic kernel parks t in HBM and re-reads the quotient: 5.6 ms for 63 000 numbers against 0.2)
    hipLaunchKernelGGL((k_div_exact_t<110, 74, 37, true>), HELPER_GRID(nb), 0, st, u, sub_small, subv, dinv, d, l, nb, count, status, flag);
    return;
  }
  if (!status && subv && wu == 220 && wl == 148 && wsub == 74) {     // digit split of the three-digit form, 2048-bit keys
    // l = (u - sub) dinv mod 2^(28 wl): a truncated product of canonical limbs (148 products a column: below 2^64) on the blocked kernel
    hipLaunchKernelGGL((k_mul_blocked<8>), dim3((unsigned)((nb + 63) / 64)), dim3(64), (size_t)wl * 256, st, u, wl, true, sub_small,
                       subv, wsub, dinv, wl, (const uint32_t*)nullptr, 0, 0u, l, wl, nb);
    return;
  }
  if (!status && (!subv || wsub == 37) && wu == 110 && wl == 74) {   // 1024-bit keys, and the halves modulo p^3 of 2048-bit keys
    hipLaunchKernelGGL((k_div_exact_nc<110, 74, 37>), HELPER_GRID(nb), 0, st, u, sub_small, subv, dinv, l, nb);
    return;
  }
  if (!status) {   // generic widths: run the check against a scratch word nobody reads (count = 0 masks every lane)
    hipLaunchKernelGGL(k_div_exact, HELPER_GRID(nb), 0, st, u, wu, sub_small, subv, wsub, tbuf, dinv, d, wd, l, wl, nb,
                       (size_t)0, (int32_t*)tbuf, 0);
    return;
  }
  hipLaunchKernelGGL(k_div_exact, HELPER_GRID(nb), 0, st, u, wu, sub_small, subv, wsub, tbuf, dinv, d, wd, l, wl, nb, count,
                     status, flag);
}
void launch_flag_not_one(const uint32_t* x, int w, size_t nb, size_t count, int32_t* status, int32_t flag, hipStream_t st) {
  hipLaunchKernelGGL(k_flag_not_one, HELPER_GRID(nb), 0, st, x, w, nb, count, status, flag);
}
void launch_is_zero(const uint32_t* x, int w, size_t nb, int32_t* flags, hipStream_t st) {
  hipLaunchKernelGGL(k_is_zero, HELPER_GRID(nb), 0, st, x, w, nb, flags);
}
void launch_select_const(const int32_t* flags, const uint32_t* c, uint32_t* x, int w, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_select_const, HELPER_GRID(nb), 0, st, flags, c, x, w, nb);
}
void launch_sub_mod(const uint32_t* a, const uint32_t* b, const uint32_t* q, uint32_t* out, int w, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_sub_mod, HELPER_GRID(nb), 0, st, a, b, q, out, w, nb);
}
void launch_copy_limbs(const uint32_t* in, int l0, int w, uint32_t* out, int wo, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_copy_limbs, HELPER_GRID(nb), 0, st, in, l0, w, out, wo, nb);
}
void launch_copy_chunks(const uint32_t* in, int w, int nchunks, uint32_t* out, size_t out_stride, int wo, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_copy_chunks, HELPER_GRID(nb), 0, st, in, w, nchunks, out, out_stride, wo, nb);
}
void launch_fill_const(const uint32_t* c, uint32_t* out, int wo, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_const, HELPER_GRID(nb), 0, st, c, out, wo, nb);
}
void launch_gather(const uint32_t* in, size_t nb_in, const uint32_t* idx, size_t n_idx, uint32_t* out, size_t nb_out, int w, hipStream_t st) {
  hipLaunchKernelGGL(k_gather, HELPER_GRID(nb_out), 0, st, in, nb_in, idx, n_idx, out, nb_out, w);
}
void launch_scatter(const uint32_t* in, size_t nb_in, const uint32_t* idx, size_t n_idx, uint32_t* out, size_t nb_out, int w, hipStream_t st) {
  hipLaunchKernelGGL(k_scatter, HELPER_GRID(n_idx ? n_idx : 1), 0, st, in, nb_in, idx, n_idx, out, nb_out, w);
}
void launch_restride(const uint32_t* in, size_t nb_in, size_t count, const uint32_t* fill, uint32_t* out, size_t nb_out, int w,
                     hipStream_t st) {
  hipLaunchKernelGGL(k_restride, HELPER_GRID(nb_out), 0, st, in, nb_in, count, fill, out, nb_out, w);
}
void launch_merge_halves(const uint32_t* lo, const uint32_t* hi, size_t half, uint32_t* out, size_t nb, int w, hipStream_t st) {
  hipLaunchKernelGGL(k_merge_halves, HELPER_GRID(2 * half), 0, st, lo, hi, half, out, nb, w);
}
void launch_sub_one(const uint32_t* x, uint32_t* out, int w, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_sub_one, HELPER_GRID(nb), 0, st, x, out, w, nb);
}
void launch_mask_bits(uint32_t* x, int w, size_t nb, size_t bits, hipStream_t st) {
  hipLaunchKernelGGL(k_mask_bits, HELPER_GRID(nb), 0, st, x, w, nb, bits);
}
void launch_sha256_transcript(const uint32_t* const* parts, const int* widths, int nparts, size_t nb, size_t count,
                              uint32_t* digest_out, int32_t* bit_out, hipStream_t st) {
  ShaArgs a;
  a.nparts = nparts;
  for (int i = 0; i < nparts && i < 6; ++i) { a.part[i].p = parts[i]; a.part[i].w = widths[i]; }
  hipLaunchKernelGGL(k_sha256_transcript, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st, a, nb, count, digest_out, bit_out);
}
void launch_equal(const uint32_t* a, const uint32_t* b, int w, size_t nb, size_t count, int32_t* ok, hipStream_t st) {
  hipLaunchKernelGGL(k_equal, HELPER_GRID(count ? count : 1), 0, st, a, b, w, nb, count, ok);
}
void launch_select(const int32_t* flags, const uint32_t* a, const uint32_t* b, uint32_t* out, int w, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_select, HELPER_GRID(nb), 0, st, flags, a, b, out, w, nb);
}
void launch_comb7_transpose(const uint32_t* mem, size_t nb, int wt, int nwin, uint32_t first, uint32_t* table, hipStream_t st) {
  const size_t total = (size_t)nwin * 128 * (size_t)wt;
  hipLaunchKernelGGL(k_comb7_transpose, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, mem, nb, wt, nwin, first, table);
}
void launch_mul_plain(const uint32_t* a, int wa, const uint32_t* b, int wb, uint32_t* out, size_t nb, uint32_t* scratch_lo, uint64_t* scratch_cy,
                      hipStream_t st) {
  // scratch_lo: (wa + wb) * nb words, scratch_cy: (wa + wb) * nb 64-bit words
  const int wo = wa + wb;
  hipLaunchKernelGGL(k_mul_plain_cols, dim3((unsigned)((nb + 255) / 256), (unsigned)wo), dim3(256), 0, st, a, wa, b, wb, a == b && wa == wb ? 1 : 0,
                     scratch_lo, scratch_cy, nb);
  hipLaunchKernelGGL(k_mul_plain_carry, HELPER_GRID(nb), 0, st, scratch_lo, scratch_cy, wo, out, nb);
}
void launch_digest_to_limbs(const uint32_t* dg, uint32_t* out, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_digest_to_limbs, HELPER_GRID(nb), 0, st, dg, out, nb);
}
void launch_unit_flags(const uint32_t* x, const uint32_t* nmod, int w, size_t nb, size_t count, uint32_t* work, int32_t* flags,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_unit_flags, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, x, nmod, w, nb, count, work, flags);
}
void launch_or_flags(const int32_t* flags, size_t count, int32_t* status, int32_t flag, hipStream_t st) {
  hipLaunchKernelGGL(k_or_flags, HELPER_GRID(count ? count : 1), 0, st, flags, count, status, flag);
}
void launch_bytes_gather_be(const uint8_t* src, const uint64_t* off, const uint32_t* len, size_t count, uint8_t* out, size_t stride,
                            hipStream_t st) {
  hipLaunchKernelGGL(k_bytes_gather_be, dim3((unsigned)(count ? count : 1)), dim3(64), 0, st, src, off, len, count, out, stride);
}
void launch_be_lengths(const uint8_t* in, size_t stride, size_t count, uint32_t* len, hipStream_t st) {
  hipLaunchKernelGGL(k_be_lengths, HELPER_GRID(count ? count : 1), 0, st, in, stride, count, len);
}
void launch_gob_emit(const uint8_t* in, size_t stride, const uint32_t* len, const uint64_t* off, size_t count, const uint8_t* prefix,
                     uint32_t prefix_len, const uint8_t* head, uint32_t head_len, const uint8_t* tail, uint32_t tail_len, uint8_t* dst,
                     hipStream_t st) {
  hipLaunchKernelGGL(k_gob_emit, dim3((unsigned)(count ? count : 1)), dim3(64), 0, st, in, stride, len, off, count, prefix, prefix_len,
                     head, head_len, tail, tail_len, dst);
}
void launch_clear_where(const int32_t* flags, size_t count, int32_t* ok, hipStream_t st) {
  hipLaunchKernelGGL(k_clear_where, HELPER_GRID(count ? count : 1), 0, st, flags, count, ok);
}
void launch_gather_rows(const uint32_t* table, int w, const int32_t* idx, size_t count, uint32_t* out, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_gather_rows, HELPER_GRID(nb), 0, st, table, w, idx, count, out, nb);
}
// Exponent reduction modulo the group order ord = 2^t m (m odd) once e mod m is known (Montgomery needs the odd part):
//   out = em + m k,  k = (e - em) m^-1 mod 2^t   (the CRT lift: out = e mod ord),
// and out += ord when that left an exponent below 3 although e itself was larger: x^e of a NON-unit x (a multiple of the
// prime) is 0 for e >= 3 and must stay so, and for units adding the order changes nothing.
__global__ void k_exp_order_lift(const uint32_t* __restrict__ e, int we, const uint32_t* __restrict__ em, int wm,
                                 const uint32_t* __restrict__ m, int t, uint32_t minv, uint32_t* __restrict__ out, int wo,
                                 size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  const uint32_t tmask = (1u << t) - 1u;
  const uint32_t k = ((e[g] - em[g]) * minv) & tmask;
  uint64_t acc = 0;
  bool small = true, same = true;
  for (int i = 0; i < wo; ++i) {
    acc += (i < wm ? em[(size_t)i * nb + g] : 0u) + (uint64_t)m[i] * k;
    const uint32_t v = (uint32_t)acc & LMASK;
    acc >>= LB;
    out[(size_t)i * nb + g] = v;
    small = small && (i == 0 ? v < 3u : v == 0u);
    same = same && (i < we ? e[(size_t)i * nb + g] == v : v == 0u);
  }
  for (int i = wo; i < we; ++i) same = same && e[(size_t)i * nb + g] == 0u;
  if (small && !same) {
    acc = 0;
    for (int i = 0; i < wo; ++i) {
      acc += out[(size_t)i * nb + g] + ((uint64_t)m[i] << t);
      out[(size_t)i * nb + g] = (uint32_t)acc & LMASK;
      acc >>= LB;
    }
  }
}
// Low limbs of two exponent expressions (all that the CRT lift over the 2-part of a group order looks at):
//   ls = x - a * e,   lb = -e      modulo 2^28, from the lowest limbs of x, a, e (limb-major arrays: limb 0 of number g at [g])
__global__ void k_exp_low_combine(const uint32_t* __restrict__ x, const uint32_t* __restrict__ a, const uint32_t* __restrict__ e,
                                  uint32_t* __restrict__ ls, uint32_t* __restrict__ lb, size_t nb) {
  CHAIN_PRIORITY();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  ls[g] = (x[g] - a[g] * e[g]) & LMASK;
  lb[g] = (0u - e[g]) & LMASK;
}
void launch_exp_low_combine(const uint32_t* x, const uint32_t* a, const uint32_t* e, uint32_t* ls, uint32_t* lb, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_exp_low_combine, HELPER_GRID(nb), 0, st, x, a, e, ls, lb, nb);
}
void launch_exp_order_lift(const uint32_t* e, int we, const uint32_t* em, int wm, const uint32_t* m, int t, uint32_t minv,
                           uint32_t* out, int wo, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_exp_order_lift, HELPER_GRID(nb), 0, st, e, we, em, wm, m, t, minv, out, wo, nb);
}

void launch_repack_windows5(const uint32_t* in, int we, uint32_t* out, int we5, size_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_repack_windows5, HELPER_GRID(nb), 0, st, in, we, out, we5, nb);
}
