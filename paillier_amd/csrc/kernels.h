// kernels.h -- launch interface between the host translation units (engine.hpp: orchestration) and kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int VM_BLOCK = 256;

// VM opcodes.  An instruction is two 32-bit words: w0 = op | (aux << 8), w1 = arg.
enum VmOp : uint32_t {
  VM_END = 0,
  VM_LOAD = 1,    // x <- mem[arg]
  VM_STORE = 2,   // mem[arg] <- x
  VM_LOADC = 3,   // x <- consts[arg]            (uniform constant)
  VM_SQR = 4,     // x <- x*x*R^-1
  VM_MUL = 5,     // x <- x*mem[arg]*R^-1
  VM_MULC = 6,    // x <- x*consts[arg]*R^-1
  VM_MULV = 7,    // x <- x*mem[aux + window(arg) of this number's exponent]*R^-1  (4-bit windows, per-number gather)
  VM_ADD = 8,     // x <- x + mem[arg]            (lazy; must be followed by a MULC before SQR)
  VM_SETOFF = 9,  // operand number offset for LOAD/STORE/MUL/ADD <- arg
  VM_MULCV = 10,  // x <- x*consts[aux + 16*arg + window(arg) of this number's exponent]*R^-1  (fixed-base comb table)
  VM_MULV5 = 11,  // MULV with 5-bit windows: `digits` holds the exponents repacked as 25-bit words (5 windows each), 32-entry table
  VM_MULV7 = 12,  // MULV with 7-bit windows: 4 per 28-bit exponent limb (no repacking), 128-entry table of NUMBER-major slots
  VM_STORET = 13, // mem[arg] <- x, number-major inside the slot ([number][WT limbs]): the layout VM_MULV7 gathers from
  VM_MULS = 14,   // mem[arg] <- x*mem[arg]*R^-1, x unchanged (a bucket of the shared chain of squarings takes the current power
                  // without the power leaving the registers; the four- and eight-lane pair kernels only)
  VM_MULVT = 15,  // VM_MULV (4-bit windows, 7 per exponent limb) on a table of NUMBER-major slots written by VM_STORET (the one-lane
                  // pair kernel for 37-limb primes and the two- and four-lane pair kernels)
  VM_MULVT5 = 16, // VM_MULV5 (5-bit windows of the repacked exponent) on a table of NUMBER-major slots; the same kernels
  VM_MULCV7 = 17, // VM_MULCV with 7-bit windows (4 per exponent limb): x <- x*consts[aux + 128*arg + window]*R^-1 (the generic kernels)
};

struct VmSeg {
  const uint32_t* prog;    // program words
  const uint32_t* nmod;    // WT modulus limbs (canonical, zero padded)
  const uint32_t* consts;  // [c][WT]
  uint32_t* mem;           // [slot][WT][nb]
  const uint32_t* digits;  // per-number exponents as 28-bit limbs, limb-major [we][nb]  (MULV)
  uint32_t n0inv;          // -N^-1 mod 2^28
  uint32_t nb;             // numbers in this segment (multiple of VM_BLOCK / K)
};

struct VmArgs {
  VmSeg seg[3];
  uint32_t seg0_blocks;  // blocks [0, seg0_blocks) run seg[0],
  uint32_t seg1_blocks;  // the next seg1_blocks run seg[1], the rest seg[2]
};

hipError_t launch_vm(int wl, int k, const VmArgs& a, uint32_t blocks, hipStream_t st);
// hand-scheduled assembly versions (asm_loader.cpp); same arguments, bit-identical results
bool vm_asm_available(int wl, int k);
// exclusive: every workgroup asks for all of a compute unit's LDS (dynamic part on top of the kernel's own), so that no other workgroup that uses LDS
// shares its CU -- one wave per SIMD for launches that run beside each other (engine.hpp exclusive_call)
// lds_share: 1 = the whole LDS of a CU (as above); 2 = more than half of it: at most ONE workgroup of such a launch per CU (other
// launches' workgroups still fit beside it) -- a launch of up to 256 workgroups then spreads over all CUs, one wave per SIMD, where the
// dispatcher sometimes stacks two of its workgroups on half the CUs (the prover's a^n | x^n launch: 28 or 47 ms, call by call)
hipError_t launch_vm_asm(int wl, int k, const VmArgs& a, uint32_t blocks, hipStream_t st, int lds_share = 0);
static_assert(sizeof(VmArgs) == 152, "VmArgs layout is hard-coded in gen_vm_asm.py (select_segment)");

void launch_unpack_be(const uint8_t* in, size_t stride, size_t nbytes, size_t count, uint32_t* out, int wt, size_t nb, hipStream_t st);
void launch_pack_be(const uint32_t* in, int wt, size_t nb, size_t count, uint8_t* out, size_t stride, size_t nbytes, hipStream_t st);
void launch_canon(uint32_t* x, const uint32_t* nmod, int wt, size_t nb, hipStream_t st);
void launch_mul_const_add(const uint32_t* a, int wa, const uint32_t* bconst, int wb, const uint32_t* addv, int wadd,
                          uint32_t add_small, uint32_t* out, int wo, size_t nb, hipStream_t st);
void launch_div_exact(const uint32_t* u, int wu, uint32_t sub_small, const uint32_t* subv, int wsub, uint32_t* tbuf,
                      const uint32_t* dinv, const uint32_t* d, int wd, uint32_t* l, int wl, size_t nb, size_t count,
                      int32_t* status, int32_t flag, hipStream_t st);
void launch_flag_not_one(const uint32_t* x, int w, size_t nb, size_t count, int32_t* status, int32_t flag, hipStream_t st);
void launch_is_zero(const uint32_t* x, int w, size_t nb, int32_t* flags, hipStream_t st);
void launch_select_const(const int32_t* flags, const uint32_t* c, uint32_t* x, int w, size_t nb, hipStream_t st);
void launch_sub_mod(const uint32_t* a, const uint32_t* b, const uint32_t* q, uint32_t* out, int w, size_t nb, hipStream_t st);
void launch_copy_limbs(const uint32_t* in, int l0, int w, uint32_t* out, int wo, size_t nb, hipStream_t st);
void launch_copy_chunks(const uint32_t* in, int w, int nchunks, uint32_t* out, size_t out_stride, int wo, size_t nb, hipStream_t st);
void launch_fill_const(const uint32_t* c, uint32_t* out, int wo, size_t nb, hipStream_t st);
void launch_gather(const uint32_t* in, size_t nb_in, const uint32_t* idx, size_t n_idx, uint32_t* out, size_t nb_out, int w, hipStream_t st);
void launch_scatter(const uint32_t* in, size_t nb_in, const uint32_t* idx, size_t n_idx, uint32_t* out, size_t nb_out, int w, hipStream_t st);
void launch_restride(const uint32_t* in, size_t nb_in, size_t count, const uint32_t* fill, uint32_t* out, size_t nb_out, int w,
                     hipStream_t st);
void launch_merge_halves(const uint32_t* lo, const uint32_t* hi, size_t half, uint32_t* out, size_t nb, int w, hipStream_t st);
void launch_sub_one(const uint32_t* x, uint32_t* out, int w, size_t nb, hipStream_t st);
void launch_mask_bits(uint32_t* x, int w, size_t nb, size_t bits, hipStream_t st);
// SHA-256 over the concatenated minimal big-endian bytes of up to 6 canonical limb-major numbers per lane
void launch_sha256_transcript(const uint32_t* const* parts, const int* widths, int nparts, size_t nb, size_t count,
                              uint32_t* digest_out, int32_t* bit_out, hipStream_t st);
void launch_equal(const uint32_t* a, const uint32_t* b, int w, size_t nb, size_t count, int32_t* ok, hipStream_t st);
void launch_select(const int32_t* flags, const uint32_t* a, const uint32_t* b, uint32_t* out, int w, size_t nb, hipStream_t st);
// comb table: entry (first + 128 i + d) of `table` ([entry][wt]) <- limbs of number i in slot 1 + d of `mem` ([slot][wt][nb]), i < nwin, d < 128
void launch_comb7_transpose(const uint32_t* mem, size_t nb, int wt, int nwin, uint32_t first, uint32_t* table, hipStream_t st);
void launch_mul_plain(const uint32_t* a, int wa, const uint32_t* b, int wb, uint32_t* out, size_t nb, uint32_t* scratch_lo, uint64_t* scratch_cy,
                      hipStream_t st);
void launch_digest_to_limbs(const uint32_t* dg, uint32_t* out, size_t nb, hipStream_t st);
// slow path of batch_inverse: flags[g] = gcd(x[g], N) != 1 (binary GCD per lane; work = 2*w*nb words of scratch)
void launch_unit_flags(const uint32_t* x, const uint32_t* nmod, int w, size_t nb, size_t count, uint32_t* work, int32_t* flags,
                       hipStream_t st);
void launch_or_flags(const int32_t* flags, size_t count, int32_t* status, int32_t flag, hipStream_t st);
void launch_clear_where(const int32_t* flags, size_t count, int32_t* ok, hipStream_t st);
// out[l][g] = table[idx[g]][l] (g < count; padding lanes take row 0)
void launch_gather_rows(const uint32_t* table, int w, const int32_t* idx, size_t count, uint32_t* out, size_t nb, hipStream_t st);
// per-number exponents as 28-bit limbs [we][nb] -> 25-bit words [we5][nb] (5 windows of 5 bits per word), for VM_MULV5
void launch_repack_windows5(const uint32_t* in, int we, uint32_t* out, int we5, size_t nb, hipStream_t st);
// out (wo limbs) = e mod 2^t m given em = e mod m (m odd, wo limbs, zero padded; minv = m^-1 mod 2^t); + the order when the
// result fell below 3 although e did not (keeps x^e = 0 for non-unit x)
// ls = x - a e, lb = -e modulo 2^28 from the lowest limbs (exponent arithmetic modulo the 2-part of a group order)
void launch_exp_low_combine(const uint32_t* x, const uint32_t* a, const uint32_t* e, uint32_t* ls, uint32_t* lb, size_t nb, hipStream_t st);
void launch_exp_order_lift(const uint32_t* e, int we, const uint32_t* em, int wm, const uint32_t* m, int t, uint32_t minv,
                           uint32_t* out, int wo, size_t nb, hipStream_t st);
// wire format (encoding/gob of Ciphertext): payload bytes between gob blobs and the fixed-stride big-endian buffers
void launch_bytes_gather_be(const uint8_t* src, const uint64_t* off, const uint32_t* len, size_t count, uint8_t* out, size_t stride,
                            hipStream_t st);
void launch_be_lengths(const uint8_t* in, size_t stride, size_t count, uint32_t* len, hipStream_t st);
void launch_gob_emit(const uint8_t* in, size_t stride, const uint32_t* len, const uint64_t* off, size_t count, const uint8_t* prefix,
                     uint32_t prefix_len, const uint8_t* head, uint32_t head_len, const uint8_t* tail, uint32_t tail_len, uint8_t* dst,
                     hipStream_t st);
