// gfx950 VALU issue-rate micro-benchmark for the instructions a big-integer
// Montgomery multiply can be built from.  Prints cycles per wave-instruction
// at 1/2/4 waves per SIMD, so DESIGN.md can state the integer-multiply roofline
// from a measurement instead of an assumption.
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// One "group" = 8 independent instructions on 8 accumulator chains, so that the
// dependent-issue latency of one chain is covered by the other seven.
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

enum Op {
  OP_MAD_U64_U32 = 0,   // v_mad_u64_u32 d, vcc = a*b + d (carry to vcc, unused)
  OP_MAD_U64_U32_ADDC,  // the pair: mad + v_addc_co_u32 consuming its carry
  OP_MAD_U64_SGPR,      // mad with one SGPR multiplicand
  OP_MUL_LO_U32,
  OP_MUL_HI_U32,
  OP_MAD_U32_U24,
  OP_MUL_HI_U32_U24,
  OP_DOT2_U32_U16,
  OP_DOT4_U32_U8,
  OP_FMA_F64,
  OP_ADD_CO_U32,
  OP_ADDC_CO_U32,
  OP_ADD_U32,
  OP_ADD3_U32,
  OP_LSHRREV_B64,
  OP_AND_B32,
  OP_MAD_I32_I24,
  OP_ADD_U32_E64,
  OP_XOR_B32,
  OP_LSHLREV_B32,
  OP_CNDMASK_B32,
  OP_ALIGNBIT_B32,
  OP_LSHL_ADD_U32,
  OP_FMA_F32,
  OP_ADD_F32,
  OP_MOV_B32,
  OP_ADD_CO_SGPR,
  OP_MAD_U64_SCARRY,
  OP_MAD_U64_DEP,
  OP_COUNT
};

static const char* op_name[OP_COUNT] = {
  "v_mad_u64_u32", "v_mad_u64_u32+v_addc_co_u32 (pair)", "v_mad_u64_u32 (sgpr src)",
  "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
  "v_dot2_u32_u16", "v_dot4_u32_u8", "v_fma_f64", "v_add_co_u32", "v_addc_co_u32",
  "v_add_u32", "v_add3_u32", "v_lshrrev_b64", "v_and_b32", "v_mad_i32_i24",
  "v_add_u32_e64", "v_xor_b32", "v_lshlrev_b32", "v_cndmask_b32", "v_alignbit_b32", "v_lshl_add_u32",
  "v_fma_f32", "v_add_f32", "v_mov_b32", "v_add_co_u32 (sgpr-pair carry)", "v_mad_u64_u32 (sgpr-pair carry)",
  "v_mad_u64_u32 (1 dependent chain)"};
static const int op_instrs_per_group[OP_COUNT] = {8, 16, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(uint64_t* out, uint32_t* sink, int iters, uint32_t seed) {
  uint32_t a = seed * (threadIdx.x + 1) | 1u, b = (seed ^ 0x9e3779b9u) * (threadIdx.x + 7) | 1u;
  uint64_t d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3;
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  uint32_t sb = __builtin_amdgcn_readfirstlane(seed | 3u);
  double f0 = a, f1 = b, f2 = 1.5, f3 = 2.5, f4 = 3.5, f5 = 4.5, f6 = 5.5, f7 = 6.5, fa = 1.0000001, fb = 1e-9;
  uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == OP_MAD_U64_U32) {
      REP16(asm volatile(
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
        "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
        "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
        "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
        : "v"(a), "v"(b) : "vcc");)
    } else if constexpr (OP == OP_MAD_U64_U32_ADDC) {
      REP16(asm volatile(
        "v_mad_u64_u32 %0, vcc, %16, %17, %0\n v_addc_co_u32 %8, vcc, 0, %8, vcc\n"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n v_addc_co_u32 %9, vcc, 0, %9, vcc\n"
        "v_mad_u64_u32 %2, vcc, %16, %17, %2\n v_addc_co_u32 %10, vcc, 0, %10, vcc\n"
        "v_mad_u64_u32 %3, vcc, %16, %17, %3\n v_addc_co_u32 %11, vcc, 0, %11, vcc\n"
        "v_mad_u64_u32 %4, vcc, %16, %17, %4\n v_addc_co_u32 %12, vcc, 0, %12, vcc\n"
        "v_mad_u64_u32 %5, vcc, %16, %17, %5\n v_addc_co_u32 %13, vcc, 0, %13, vcc\n"
        "v_mad_u64_u32 %6, vcc, %16, %17, %6\n v_addc_co_u32 %14, vcc, 0, %14, vcc\n"
        "v_mad_u64_u32 %7, vcc, %16, %17, %7\n v_addc_co_u32 %15, vcc, 0, %15, vcc\n"
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7),
          "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a), "v"(b) : "vcc");)
    } else if constexpr (OP == OP_MAD_U64_SGPR) {
      REP16(asm volatile(
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
        "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
        "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
        "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
        : "v"(a), "s"(sb) : "vcc");)
    } else if constexpr (OP == OP_MAD_U64_SCARRY) {
      REP16(asm volatile(
        "v_mad_u64_u32 %0, s[40:41], %8, %9, %0\n v_mad_u64_u32 %1, s[42:43], %8, %9, %1\n"
        "v_mad_u64_u32 %2, s[40:41], %8, %9, %2\n v_mad_u64_u32 %3, s[42:43], %8, %9, %3\n"
        "v_mad_u64_u32 %4, s[40:41], %8, %9, %4\n v_mad_u64_u32 %5, s[42:43], %8, %9, %5\n"
        "v_mad_u64_u32 %6, s[40:41], %8, %9, %6\n v_mad_u64_u32 %7, s[42:43], %8, %9, %7\n"
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
        : "v"(a), "v"(b) : "s40", "s41", "s42", "s43");)
    } else if constexpr (OP == OP_MAD_U64_DEP) {
      REP16(asm volatile(
        "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
        "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
        "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
        "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
        : "+v"(d0) : "v"(a), "v"(b) : "vcc");)
    } else if constexpr (OP == OP_FMA_F64) {
      REP16(asm volatile(
        "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
        "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
        : "v"(fa), "v"(fb));)
    } else if constexpr (OP == OP_LSHRREV_B64) {
      REP16(asm volatile(
        "v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3\n"
        "v_lshrrev_b64 %4, 1, %4\n v_lshrrev_b64 %5, 1, %5\n v_lshrrev_b64 %6, 1, %6\n v_lshrrev_b64 %7, 1, %7\n"
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
    } else {
      // 32-bit destination ops: c_k = op(a, b, c_k)
#define G32(STR) REP16(asm volatile( \
        STR(%0) STR(%1) STR(%2) STR(%3) STR(%4) STR(%5) STR(%6) STR(%7) \
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
        : "v"(a), "v"(b) : "vcc");)
#define G32S(STR) REP16(asm volatile( \
        STR(%0) STR(%1) STR(%2) STR(%3) STR(%4) STR(%5) STR(%6) STR(%7) \
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
        : "v"(a), "v"(b) : "s40", "s41");)
#define S_MUL_LO(r) "v_mul_lo_u32 " #r ", %8, " #r "\n"
#define S_MUL_HI(r) "v_mul_hi_u32 " #r ", %8, " #r "\n"
#define S_MAD24(r) "v_mad_u32_u24 " #r ", %8, %9, " #r "\n"
#define S_MADI24(r) "v_mad_i32_i24 " #r ", %8, %9, " #r "\n"
#define S_MULHI24(r) "v_mul_hi_u32_u24 " #r ", %8, " #r "\n"
#define S_DOT2(r) "v_dot2_u32_u16 " #r ", %8, %9, " #r "\n"
#define S_DOT4(r) "v_dot4_u32_u8 " #r ", %8, %9, " #r "\n"
#define S_ADDCO(r) "v_add_co_u32 " #r ", vcc, %8, " #r "\n"
#define S_ADDC(r) "v_addc_co_u32 " #r ", vcc, %8, " #r ", vcc\n"
#define S_ADD(r) "v_add_u32 " #r ", %8, " #r "\n"
#define S_ADD3(r) "v_add3_u32 " #r ", %8, %9, " #r "\n"
#define S_AND(r) "v_and_b32 " #r ", %8, " #r "\n"
#define S_ADD64(r) "v_add_u32_e64 " #r ", %8, " #r "\n"
#define S_XOR(r) "v_xor_b32 " #r ", %8, " #r "\n"
#define S_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define S_CND(r) "v_cndmask_b32 " #r ", %8, " #r ", vcc\n"
#define S_ALIGN(r) "v_alignbit_b32 " #r ", %8, " #r ", 7\n"
#define S_LSHLADD(r) "v_lshl_add_u32 " #r ", %8, 3, " #r "\n"
#define S_FMA32(r) "v_fma_f32 " #r ", %8, %9, " #r "\n"
#define S_ADDF32(r) "v_add_f32 " #r ", %8, " #r "\n"
#define S_MOV(r) "v_mov_b32 " #r ", %8\n"
#define S_ADDCOS(r) "v_add_co_u32 " #r ", s[40:41], %8, " #r "\n"
      if constexpr (OP == OP_MUL_LO_U32) { G32(S_MUL_LO) }
      else if constexpr (OP == OP_MUL_HI_U32) { G32(S_MUL_HI) }
      else if constexpr (OP == OP_MAD_U32_U24) { G32(S_MAD24) }
      else if constexpr (OP == OP_MAD_I32_I24) { G32(S_MADI24) }
      else if constexpr (OP == OP_MUL_HI_U32_U24) { G32(S_MULHI24) }
      else if constexpr (OP == OP_DOT2_U32_U16) { G32(S_DOT2) }
      else if constexpr (OP == OP_DOT4_U32_U8) { G32(S_DOT4) }
      else if constexpr (OP == OP_ADD_CO_U32) { G32(S_ADDCO) }
      else if constexpr (OP == OP_ADDC_CO_U32) { G32(S_ADDC) }
      else if constexpr (OP == OP_ADD_U32) { G32(S_ADD) }
      else if constexpr (OP == OP_ADD3_U32) { G32(S_ADD3) }
      else if constexpr (OP == OP_AND_B32) { G32(S_AND) }
      else if constexpr (OP == OP_ADD_U32_E64) { G32(S_ADD64) }
      else if constexpr (OP == OP_XOR_B32) { G32(S_XOR) }
      else if constexpr (OP == OP_LSHLREV_B32) { G32(S_LSHL) }
      else if constexpr (OP == OP_CNDMASK_B32) { G32(S_CND) }
      else if constexpr (OP == OP_ALIGNBIT_B32) { G32(S_ALIGN) }
      else if constexpr (OP == OP_LSHL_ADD_U32) { G32(S_LSHLADD) }
      else if constexpr (OP == OP_FMA_F32) { G32(S_FMA32) }
      else if constexpr (OP == OP_ADD_F32) { G32(S_ADDF32) }
      else if constexpr (OP == OP_MOV_B32) { G32(S_MOV) }
      else if constexpr (OP == OP_ADD_CO_SGPR) { G32S(S_ADDCOS) }
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t r = (uint32_t)(d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7) ^ (uint32_t)((d0 ^ d4) >> 32) ^
               c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
  if (r == 0x12345678u) sink[0] = r;  // keep everything live
  if ((threadIdx.x & 63) == 0) { size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; }
}

template <int OP>
static void run_op(uint64_t* d_out, uint32_t* d_sink, int n_cu, double clk_ghz_hint) {
  const int iters = 4000;
  const int groups_per_iter = 16;
  for (int waves_per_simd : {1, 2, 4}) {
    int blocks = n_cu * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    size_t n_waves = (size_t)blocks * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<OP><<<blocks, 256>>>(d_out, d_sink, 10, 12345u);  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, 256>>>(d_out, d_sink, iters, 12345u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(2 * n_waves);
    CK(hipMemcpy(h.data(), d_out, 2 * n_waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    double sum = 0, rsum = 0; for (size_t i = 0; i < n_waves; ++i) { sum += (double)h[2 * i]; rsum += (double)h[2 * i + 1]; }
    double mt_ghz = sum / rsum * 0.1;  // memtime ticks per 100 MHz realtime tick
    double cyc_per_wave = sum / n_waves;
    double instrs = (double)iters * groups_per_iter * op_instrs_per_group[OP];
    double cyc_per_instr_wave = cyc_per_wave / instrs;                 // as seen by one wave
    double cyc_per_instr_simd = cyc_per_instr_wave / waves_per_simd;   // SIMD issue cost
    double total_instr = instrs * n_waves;
    double ginstr_s = total_instr / (ms * 1e-3) / 1e9;                 // wave-instructions/s
    printf("%-40s waves/SIMD=%d  cyc/instr(wave)=%6.2f  cyc/instr(SIMD)=%6.2f  wall=%7.3f ms  "
           "%8.2f Gwave-instr/s  lane-ops=%8.3f Tops/s  memtime=%5.3f GHz  wall-cyc/instr(SIMD)@2.4GHz=%5.2f\n",
           op_name[OP], waves_per_simd, cyc_per_instr_wave, cyc_per_instr_simd, ms, ginstr_s,
           ginstr_s * 64 / 1e3, mt_ghz, 2.4e9 / (ginstr_s * 1e9 / (n_cu * 4)));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
  (void)clk_ghz_hint;
}

template <int OP>
static void run_all(uint64_t* d_out, uint32_t* d_sink, int n_cu, double clk) {
  run_op<OP>(d_out, d_sink, n_cu, clk);
  if constexpr (OP + 1 < OP_COUNT) run_all<OP + 1>(d_out, d_sink, n_cu, clk);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  arch=%s  CUs=%d  clock=%d kHz  memclk=%d kHz\n", p.name, p.gcnArchName,
         p.multiProcessorCount, p.clockRate, p.memoryClockRate);
  int n_cu = p.multiProcessorCount;
  uint64_t* d_out; uint32_t* d_sink;
  CK(hipMalloc(&d_out, sizeof(uint64_t) * n_cu * 4 * 4 * 4 * 2));
  CK(hipMalloc(&d_sink, 64));
  run_all<0>(d_out, d_sink, n_cu, p.clockRate * 1e-6);
  return 0;
}
