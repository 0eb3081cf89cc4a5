// Issue cost of the instructions that sit beside the multiplies in the rows of the lane-sliced kernels (gfx950), one and two
// waves per SIMD: independent streams over 8 register groups, s_memtime around 4000 x 128 instructions.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rates2.hip -o /tmp/valu2 && /tmp/valu2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

enum Op { MAD_U64, MAD_I64, MAD_I64_INL, LSHL_ADD_U64, LSHRREV_B64, LSHLREV_B64, AND_DPP, MOV_DPP, MUL_LO, BFI, AND_LIT, SUBB_PAIR, OPS };
static const char* names[OPS] = {"v_mad_u64_u32", "v_mad_i64_i32 (vgpr mask)", "v_mad_i64_i32 (inline -1)", "v_lshl_add_u64", "v_lshrrev_b64 28",
                                 "v_lshlrev_b64 (vgpr shift)", "v_and_b32_dpp quad_perm", "v_mov_b32_dpp quad_perm", "v_mul_lo_u32 (sgpr)",
                                 "v_bfi_b32", "v_and_b32 literal", "v_sub_co_u32 + v_subb_co_u32 (pair)"};

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t* sink, int iters) {
  uint64_t d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
  uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x5bd1e995u, m = (threadIdx.x & 1) ? 0xffffffffu : 0u;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define G64(S) REP16(asm volatile(S(%0) S(%1) S(%2) S(%3) S(%4) S(%5) S(%6) S(%7) \
      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b), "v"(m) : "vcc");)
#define S_MADU(r) "v_mad_u64_u32 " #r ", vcc, %8, %9, " #r "\n"
#define S_MADI(r) "v_mad_i64_i32 " #r ", vcc, %8, %10, " #r "\n"
#define S_MADII(r) "v_mad_i64_i32 " #r ", vcc, %8, -1, " #r "\n"
#define S_LSHLADD(r) "v_lshl_add_u64 " #r ", " #r ", 0, %8\n"
#define S_LSHR(r) "v_lshrrev_b64 " #r ", 28, " #r "\n"
#define S_LSHL(r) "v_lshlrev_b64 " #r ", %10, " #r "\n"
    if constexpr (OP == MAD_U64) { G64(S_MADU) }
    else if constexpr (OP == MAD_I64) { G64(S_MADI) }
    else if constexpr (OP == MAD_I64_INL) { G64(S_MADII) }
    else if constexpr (OP == LSHRREV_B64) { G64(S_LSHR) }
    else if constexpr (OP == LSHLREV_B64) { G64(S_LSHL) }
    else if constexpr (OP == LSHL_ADD_U64) {
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                         "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(d0 ^ d1));)
    } else {
      uint32_t c0 = (uint32_t)d0, c1 = (uint32_t)d1, c2 = (uint32_t)d2, c3 = (uint32_t)d3, c4 = (uint32_t)d4, c5 = (uint32_t)d5, c6 = (uint32_t)d6, c7 = (uint32_t)d7;
#define G32(S) REP16(asm volatile(S(%0) S(%1) S(%2) S(%3) S(%4) S(%5) S(%6) S(%7) \
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b), "s"(0x1234567) : "vcc");)
#define S_ANDDPP(r) "v_and_b32_dpp " #r ", " #r ", %8 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n"
#define S_MOVDPP(r) "v_mov_b32_dpp " #r ", " #r " quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n"
#define S_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %10\n"
#define S_BFI(r) "v_bfi_b32 " #r ", %8, %9, " #r "\n"
#define S_ANDLIT(r) "v_and_b32 " #r ", 0xfffffff, " #r "\n"
#define S_SUBB(r) "v_sub_co_u32 " #r ", vcc, " #r ", %8\n v_subb_co_u32 " #r ", vcc, " #r ", %9, vcc\n"
      if constexpr (OP == AND_DPP) { G32(S_ANDDPP) }
      else if constexpr (OP == MOV_DPP) { G32(S_MOVDPP) }
      else if constexpr (OP == MUL_LO) { G32(S_MULLO) }
      else if constexpr (OP == BFI) { G32(S_BFI) }
      else if constexpr (OP == AND_LIT) { G32(S_ANDLIT) }
      else if constexpr (OP == SUBB_PAIR) { G32(S_SUBB) }
      d0 ^= c0; d1 ^= c1; d2 ^= c2; d3 ^= c3; d4 ^= c4; d5 ^= c5; d6 ^= c6; d7 ^= c7;
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t r = (uint32_t)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7) ^ (uint32_t)((d0 ^ d5) >> 32);
  if (r == 0x12345678u) sink[0] = r;
  if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
static int run(uint64_t* d_out, uint32_t* d_sink, int n_cu) {
  const int iters = 4000;
  for (int wps : {1, 2}) {
    int blocks = n_cu * wps;
    k<OP><<<blocks, 256>>>(d_out, d_sink, 8);
    CK(hipDeviceSynchronize());
    k<OP><<<blocks, 256>>>(d_out, d_sink, iters);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h((size_t)blocks * 4);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double ticks = (double)h[h.size() / 2];                 // s_memtime ticks at 100 MHz on this part: convert with the shader clock below
    const double insts = (double)iters * 128 * (OP == SUBB_PAIR ? 2 : 1);
    printf("%-40s %d wave(s)/SIMD: %.3f memtime ticks per wave-instruction (x %d waves)\n", names[OP], wps, ticks / insts, wps);
  }
  return 0;
}

int main() {
  int n_cu = 256;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); n_cu = p.multiProcessorCount;
  uint64_t* d_out; uint32_t* d_sink;
  CK(hipMalloc(&d_out, (size_t)n_cu * 2 * 4 * 8)); CK(hipMalloc(&d_sink, 64));
  printf("CUs %d; ratios to the first line are what matters (s_memtime runs at a fixed 100 MHz)\n", n_cu);
  if (run<MAD_U64>(d_out, d_sink, n_cu)) return 1;
  if (run<MAD_I64>(d_out, d_sink, n_cu)) return 1;
  if (run<MAD_I64_INL>(d_out, d_sink, n_cu)) return 1;
  if (run<LSHL_ADD_U64>(d_out, d_sink, n_cu)) return 1;
  if (run<LSHRREV_B64>(d_out, d_sink, n_cu)) return 1;
  if (run<LSHLREV_B64>(d_out, d_sink, n_cu)) return 1;
  if (run<AND_DPP>(d_out, d_sink, n_cu)) return 1;
  if (run<MOV_DPP>(d_out, d_sink, n_cu)) return 1;
  if (run<MUL_LO>(d_out, d_sink, n_cu)) return 1;
  if (run<BFI>(d_out, d_sink, n_cu)) return 1;
  if (run<AND_LIT>(d_out, d_sink, n_cu)) return 1;
  if (run<SUBB_PAIR>(d_out, d_sink, n_cu)) return 1;
  return 0;
}
