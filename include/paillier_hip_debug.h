/* paillier_hip_debug.h -- test hooks of libpaillier_hip.so.  NOT part of the drop-in boundary (that is paillier_hip.h): nothing
 * here replaces a reference call site, a Go / C caller never needs it, and it may change without notice.  Included by the
 * library's own debug.cpp, by tests/ and tools/ only.
 *
 * Environment variables the library reads -- two, both diagnostics that print to stderr and change nothing that is launched:
 *   PGPU_PROFILE_DUMP=1   pgpu_ctx_last_profile prints one line per profiled VM launch of the last call (kernel, ms, multiply-adds,
 *                         fraction of the issue peak); the prover prints how many instances drew challenge bit 1
 *   PGPU_HOST_TRACE=1     the prover prints host-side time stamps of its stages
 * Everything that changes the plan is a pgpu_ctx_set_flag name (paillier_hip.h); tests/test_plan_cpu.py enforces both statements.
 */
#ifndef PAILLIER_HIP_DEBUG_H
#define PAILLIER_HIP_DEBUG_H

#include "paillier_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* Runs a raw VM program (kernels.h opcodes; pairs of words) on raw slot memory: host array of 28-bit limbs,
 * limb-major [slot][WT][nb], nb a multiple of 256.  use_asm selects the assembly or the hipcc kernel.
 * For tests only: lets the two implementations of the VM be compared opcode by opcode. */
int pgpu_vm_debug_run(const pgpu_modulus* mod, const uint32_t* prog, size_t prog_words, uint32_t* mem_host,
                      size_t nslots, size_t nb, int use_asm, int* wt_out);

/* Same for the pair kernel (residues modulo p^2 as two base-p digits, the Decrypt ladder of 2048-bit keys): slots are
 * [2H][nb] limb arrays (digit a0 in limbs 0..H-1, a1 in limbs H..2H-1), SQR / MUL are the pair operations.
 * lanes = 1: the one-lane kernel for N = p^2 with a 37-limb prime (Decrypt); lanes = 2: the two-lane kernel for N = n^2 with
 * a 74-limb public n (Encrypt, PartialDecrypt, ...; any odd n).
 * consts_out (optional, 2H words) receives the kernel's constants p | Cadj, h_out the digit width H. */
int pgpu_pair_debug_run(pgpu_ctx* ctx, const uint8_t* p_be, size_t p_len, int lanes, const uint32_t* prog, size_t prog_words,
                        uint32_t* mem_host, size_t nslots, size_t nb, uint32_t* consts_out, int* h_out);

/* The planning predicates of paillier_amd/csrc/plan.hpp -- which kernel shape, how many lanes per number, which window width,
 * whether a ladder is split -- as one C entry point, so that they can be unit-tested without a GPU (tests/test_plan_cpu.py).
 * The protocol bodies call the same functions.  `what` names the decision, args its inputs, out its outputs:
 *   "triple_window_bits"  (nb, H)                                  -> win
 *   "crt3_ladder"         (nb, H, wt3, lanes_wanted, per_number, prereq) -> triple, win, nm5, split
 *   "crt3_two"            (nb, H, lanes_wanted, prereq)            -> usable, split
 *   "early_response_ok"   (nb_instances, H)                        -> ok
 *   "response_by_structure" (statements, instances, nb_instances, lanes_wanted) -> ok
 *   "extract_beside"      (nb_statements, nb_instances, lanes_wanted) -> ok
 *   "shared_chain_groups" (nb_ciphertexts, n_shares, lanes_wanted, have_eight_lane_kernel) -> groups of shares with a chain of squarings each
 *   "lanes_target"        (lanes_wanted, stream_cus) -> lanes that fill what the context's stream may use with one wave per SIMD
 *   "lds_share"           (blocks, stream_cus, on_side, in_exclusive_call, products, exclusive_flag, spread_flag) -> 0 | 1 | 2
 *                         (LDS a workgroup asks for: the kernel's own | a whole CU's | just over half -- placement by LDS size)
 *   "generic_shape"       (WL, K, launch_nb, segments, lanes_wanted, wave_sliced_ok) -> WL, K   (lanes per number of the generic kernels)
 *   "pair_lanes_shared"   (numbers, lanes_wanted, have4, have8)    -> lanes
 *   "pair_lanes_2or4"     (numbers, lanes_wanted, have4)           -> lanes
 *   "crt_pair_lanes"      (key_lanes, have_two_lane_variant, nb, lanes_wanted) -> lanes, usable (for 37-limb primes)
 *   "dual_pair_window_bits" (nb, w2, nm)                           -> wb (0: not on the pair kernels)
 *   "pair_nm4_fits"       (nb, w2)                                 -> ok
 *   "shared_chain_pays"   (nb, lanes_wanted)                       -> ok
 *   "triple_two_lanes_per_digit" (nb, lanes_wanted)                -> ok
 *   "dual_n3_two_ladders" (nb, lanes_wanted) -> ok   (x^(e0) and W^n modulo n^3 as two eight-lane ladders side by side)
 *   "perlane_table_slots" (wb, nm)                                 -> slots
 *   "gather_entries"      (wb)                                     -> entries
 * lanes_wanted = 0 means the default (one wave on every SIMD: 65 536 lanes).  Returns the number of outputs written, or
 * PGPU_ERR_INVALID for an unknown name / too few arguments / too small an output array.  Needs no context and no GPU. */
int pgpu_plan_query(const char* what, const uint64_t* args, int nargs, int64_t* out, int nout);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* PAILLIER_HIP_DEBUG_H */
