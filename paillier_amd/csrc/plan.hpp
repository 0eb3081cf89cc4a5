// plan.hpp -- every SIZE decision of the engine in one place: which kernel shape, how many lanes per number, which window
// width, whether a ladder is split.  Pure host functions of (batch size, limb counts, occupancy target, switches) -- no HIP, no
// context, no key material -- so that they can be unit-tested without a GPU (tests/test_plan_cpu.py through pgpu_plan_query of
// include/paillier_hip_debug.h) and so that two call sites cannot disagree about the same bound.  (Round 3 shipped a regression
// exactly there: the split gate of the prover's Alpha ladder bounded 128 + 64 table slots where the window-width choice bounded
// the 128 entries the gathers address; between 50 121 and 74 986 numbers the p-adic split was silently dropped.)
//
// Vocabulary: nb = numbers of a launch (padded to 256); H = limbs of a digit (37 for the primes of a 2048-bit key, 74 for its n);
// lt = lanes that fill the chip with one wave per SIMD (1024 SIMDs x 64), or the caller's override (pgpu_ctx_set_flag
// "lanes_wanted").
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace plan {

constexpr int LB = 28;                          // bits of a limb
constexpr uint32_t kChipCUs = 256;
constexpr size_t kChipLanes = (size_t)1024 * 64;   // 256 CUs x 4 SIMDs x 64 lanes: one wave on every SIMD
constexpr uint64_t kGatherSpan = 1ull << 32;    // per-number table gathers of the assembly kernels use 32-bit byte offsets

// stream_cus: compute units the context's stream may use: a context with a CU partition fills ITS slice.  Eight contexts on 32 CUs each,
// a 2 048-ciphertext PartialDecrypt in flight on every one (tools/small_batch_sweep.py, profiles/r05_small_batch_overlap.jsonl): four lanes
// per number -- one wave per SIMD of the slice -- where the whole chip's target picked eight lanes and two waves per SIMD: 72.2 -> 62.1 ms
// for the eight calls; 4 096 ciphertexts each: 145.5 -> 120.0 ms.
inline size_t lanes_target(size_t ctx_lanes_wanted, uint32_t stream_cus = kChipCUs) {
  return ctx_lanes_wanted ? ctx_lanes_wanted : (size_t)stream_cus * 4 * 64;
}

// ---- per-number window tables -------------------------------------------------------------------------------------------------
// windows of a `we`-limb exponent: 4 bits = 7 per 28-bit limb (VM_MULV / VM_MULVT), 7 bits = 4 per limb (VM_MULV7), 5 bits = 5
// per repacked 25-bit word (VM_MULV5 / VM_MULVT5)
inline int perlane_windows(int we, int wb) { return wb == 4 ? we * 7 : wb == 5 ? (we * LB + 4) / 5 : we * 4; }
// slots a table OCCUPIES: the 2^wb gathered entries and, for number-major tables (nm), the limb-major copies of the entries the
// table build itself reads back (7 bits: 64; 5 bits: 16; 4 bits: x alone)
inline int perlane_table_slots(int wb, bool nm = false) { return wb == 7 ? 128 + 64 : nm ? (wb == 5 ? 32 + 16 : 16 + 1) : 1 << wb; }
// entries a GATHER can address (+ 1: the lane's own offset inside the last entry).  Only these must lie within the 32-bit
// offsets -- the limb-major scratch copies behind a number-major table are reached through 64-bit slot bases.
inline int gather_entries(int wb) { return (1 << wb) + 1; }
inline bool gather_fits(size_t nb, int slot_limbs, int entries) { return (uint64_t)nb * (uint64_t)slot_limbs * 4u * (uint64_t)entries < kGatherSpan; }

// sliding-window width of the SHARED exponent that rides on the same chain of squarings as per-number windows of wb bits
inline int dual_sliding_bits(int wb) { return wb == 7 ? 7 : 6; }

// three-digit kernels (slots of 3H limbs): 7-bit per-number windows while the 128 gathered entries stay inside the offsets
// (75 000 numbers for 37-limb digits, 37 000 for 74-limb digits), else 5-bit windows (number-major as well: VM_MULVT5)
inline int triple_window_bits(size_t nb, int H) { return gather_fits(nb, 3 * H, gather_entries(7)) ? 7 : 5; }
// the three-digit kernel serves a batch at all (its widest unconditional gather: 32 limb-major entries of a WT(n^3)-limb slot)
inline bool triple_batch_fits(size_t nb, int wt3) { return gather_fits(nb, wt3 + 4, 33); }

// pair kernels, per-number windows: number-major 4-bit tables (VM_MULVT) while 16 entries + x fit
inline bool pair_nm4_fits(size_t nb, int w2) { return gather_fits(nb, w2, 18); }
// ... limb-major 4-bit tables (VM_MULV)
inline bool pair_mulv_fits(size_t nb, int w2) { return gather_fits(nb, w2, 17); }
// interleaved ladder modulo n^2 (dual_pow_pair): 5-bit windows while the table fits, else 4-bit, else not on the pair kernels (0)
inline int dual_pair_window_bits(size_t nb, int w2, bool nm) {
  if (gather_fits(nb, w2, perlane_table_slots(5, nm) + 1)) return 5;
  if (gather_fits(nb, w2, perlane_table_slots(4, nm) + 1)) return 4;
  return 0;
}

// ---- lanes per number -----------------------------------------------------------------------------------------------------------
// Shared-exponent ladder modulo n^2 in pair form (GenQ 2 lanes / GenQ4 4 lanes / GenQ8 8 lanes per number).  `numbers` = all the
// numbers of the launch (segments included).  Two lanes from one wave per SIMD upwards; below that four (a squaring is half as
// long: the ladder's latency is the run time); eight while every wave still has a SIMD of its own.  0: no pair kernel fits.
// (round 5: sixteen -- GenQ16, a DPP row per number: a squaring is 80 rows of ~34 instructions where the eight-lane kernel has 76 rows of
// 51, but the quotient chain of a row -- six dependent steps, one multiply apart -- is no longer hidden: 184 cycles a row measured, 6.4 us
// a squaring against 7.4.  Up to 2 048 numbers: callers run TWO such ladders side by side (the verifier's E^n and y^n), and at 4 096
// numbers each the pair took 25.9 ms where the eight-lane kernel takes 18)
inline int pair_lanes_shared(size_t numbers, size_t lt, bool have4, bool have8, bool have16 = false) {
  if (numbers * 2 >= lt) return 2;
  if (!have4) return 2;
  if (have8 && have16 && numbers * 16 * 2 <= lt) return 16;
  if (have8 && numbers * 8 <= lt) return 8;
  return 4;
}
// a batch below one wave per SIMD at two lanes per number needs the four-lane kernel; a key whose digits it does not serve
// (odd limb counts) leaves such batches to the other kernel families
inline bool pair_kernel_serves(size_t numbers, size_t lt, bool have4) { return numbers * 2 >= lt || have4; }
// ... without the eight-lane kernel in play (per-number windows, program segments that share a launch)
inline int pair_lanes_2or4(size_t numbers, size_t lt, bool have4) { return (numbers * 2 >= lt || !have4) ? 2 : 4; }

// CRT halves modulo p^2, q^2 on the one-lane pair kernel (GenP, 37-limb primes): two lanes per number only while they leave every
// wave a SIMD of its own (both halves: 4 nb lanes); between half a wave and one wave per SIMD at one lane, two lanes would put
// two waves on most SIMDs (16.9 against 14.5 ms for 20 480 ... 30 720 ciphertexts)
inline int crt_pair_lanes(int key_lanes, bool have_two_lane_variant, size_t nb, size_t lt) {
  return (key_lanes == 1 && have_two_lane_variant && nb * 4 <= lt) ? 2 : key_lanes;
}
// ... and EIGHT lanes per number (vm_asm_10_96: the digits of 37-limb primes in four slices of 10 limbs, a squaring 40 rows of ~30
// instructions where the two-lane kernel has 37 rows of ~84) while both halves fit two waves per SIMD: up to 8 192 numbers.  The latency
// of the ladder is the run time there -- Decrypt-2048 of up to 4 096 ciphertexts 8.8 -> 4.0 ms per call, 8 192: 8.9 -> 6.3 (two waves per
// SIMD), 12 288: 8.9 | 8.7; a rank's 2 048 prover instances: a^n | x^n's second stage 9.0 -> 4.9 ms.  Inside a call whose launches run
// beside each other (`beside` 2: the DDLEQ prover) only while one wave per SIMD suffices: at 4 096 instances the launch took 6.2 instead
// of 9.8 ms and the call 53.7 instead of 51.9 -- both wave slots of every SIMD were taken from ct1's decryption for that long
inline bool crt_pair_lanes8(size_t nb, size_t lt, bool have8, bool enabled, int beside = 1) {
  return have8 && enabled && nb * 2 * 8 * (size_t)beside <= 2 * lt;
}
// ... and the ladders modulo p^3, q^3 (level-two Decrypt, the prover's lifts and plaintexts) with TWO lanes per digit (vm_asm_19_112: the
// digits of 37-limb primes in two slices of 19 limbs; 38 rows of ~55 instructions where one lane per digit has 37 rows of ~93) while both
// halves leave every wave a SIMD of its own: up to 4 096 numbers, counted with the launches known to run beside this one
inline bool crt_triple_lanes6(size_t nb, size_t lt, bool have6, bool enabled, int beside = 1) {
  return have6 && enabled && nb * 2 * 8 * (size_t)beside <= lt;
}
// whether the CRT halves of Decrypt take the pair kernels at all at this batch size
inline bool crt_pair_usable(int lanes_now, int prime_limbs, size_t nb, size_t lt) { return lanes_now == 1 || prime_limbs <= 55 || nb * 4 >= lt; }

// ladders modulo the 37-limb PRIMES of a 2048-bit key (both halves in one launch: 2 nb numbers): one lane per number on the unrolled
// kernel (2 053 multiplies in a row per squaring, ~2 300 issue slots) -- or four lanes of 10 limbs (vm_asm_10_4, GenS4: 40 rows of 27
// instructions, ~1 100 slots per lane) while that still leaves every wave a SIMD of its own: what counts below a quarter of a wave per
// SIMD is the latency of one ladder (2 048 prover instances: 5.6 -> 3.1, 7.3 -> 3.7, 7.7 -> 3.9 ms for the three ladders of the
// critical path; 8 192 numbers of a^n | x^n at 4 096 instances: 6.0 -> 3.3).  Four lanes cost twice the issue slots of one, so a wave
// that has to SHARE its SIMD gains nothing (X modulo the primes of 4 096 instances, 8 192 numbers beside ct1's decryption: 7.3 -> 9.7
// ms): `beside` = the launches known to run next to this one, the ladder included (1: alone; s on a side lane: 4).
inline int prime_lanes(size_t nb, size_t lt, bool have_sliced, bool enabled, int beside = 1) {
  return (have_sliced && enabled && nb * 2 * 4 * (size_t)beside <= lt) ? 4 : 1;
}

// three-digit kernel: two lanes per digit (GenQ6) for batches so small that eight lanes per number still leave every wave a SIMD
inline bool triple_two_lanes_per_digit(size_t nb, size_t lt) { return nb * 8 <= lt; }
// ... four lanes per digit (GenQ12, a DPP row per number) where the two ladders of dual_n3_two_ladders still find a SIMD per wave at sixteen
// lanes per number each: up to 2 048 numbers
// (`ladders`: how many such launches run side by side -- the verifier's two; a lone ladder, level-two Encrypt's (r^n)^n, up to 4 096)
inline bool triple_four_lanes_per_digit(size_t nb, size_t lt, bool have12, bool enabled, int ladders = 2) {
  return have12 && enabled && nb * 16 * (size_t)ladders <= lt;
}
// ... and x^(e0) W^n modulo n^3 (the verifier, NestedRandomize) as TWO such ladders side by side instead of one interleaved chain while
// both still find a SIMD per wave: up to 4 096 numbers
inline bool dual_n3_two_ladders(size_t nb, size_t lt) { return nb * 8 * 2 <= lt; }

// several shares on the same ciphertexts: ONE chain of squarings when the batch fills at least half the chip on its own
inline bool shared_chain_pays(size_t nb, size_t lt) { return nb * 8 >= lt; }

// ---- the key holder's ladders modulo p^3, q^3 (DDLEQ prover) -------------------------------------------------------------------
struct Crt3Ladder {
  bool triple;     // both halves on the three-digit kernel
  int win;         // per-number window bits there (7 / 5)
  bool nm5;        // 5-bit windows on number-major tables
  bool split;      // p-adic split: a stage of 2 047 squarings modulo p^2 on the pair kernel, then 1 024 modulo p^3
};
// per_number: the ladder has per-number exponents (Alpha / sanity / response); prereq: what the split needs from the key and the
// switches (lift on, exponents reduced modulo the group orders, the pair kernel for p^2 ...), decided by the caller
inline Crt3Ladder crt3_ladder(size_t nb, int H, int wt3, size_t lt, bool per_number, bool prereq) {
  Crt3Ladder l{};
  l.triple = triple_batch_fits(nb, wt3);
  l.win = per_number ? triple_window_bits(nb, H) : 5;
  l.nm5 = per_number && l.win == 5;
  // below one wave per SIMD for the stage modulo p^2 a ladder's length binds and one ladder is shorter than two; the windows of
  // r0 on the digit kernel must fit the gather offsets -- the ENTRIES the gathers address (gather_entries), not the slots the
  // table occupies
  l.split = l.triple && prereq && per_number && nb * 4 >= lt && gather_fits(nb, 3 * H, gather_entries(l.win));
  return l;
}
// the response ladder with two per-number exponents (pow_n3_crt_two): 7-bit windows only; its own p-adic split likewise
struct Crt3Two { bool usable; bool split; };
inline Crt3Two crt3_two(size_t nb, int H, size_t lt, bool prereq) {
  Crt3Two t{};
  t.usable = triple_window_bits(nb, H) == 7;
  t.split = t.usable && prereq && nb * 4 >= lt && gather_fits(nb, 3 * H, gather_entries(7));
  return t;
}
// the response may be prepared for EVERY instance before the challenge bits are known only where its kernels are certain
// whatever the number of bit-1 instances turns out to be (all of them at worst)
inline bool early_response_ok(size_t nb_instances, int H) { return triple_window_bits(nb_instances, H) == 7; }
// the response through the structure of the unit group costs two level-two decryptions PER STATEMENT (of s and of b) and saves
// more than half of every bit-1 instance's ladder: from four instances per statement
// ... and for batches so small that the chip is mostly idle (up to 4 096 instances; at 8 192 it measured slower: 101 against
// 96 ms): there the extra decryptions run beside the other launches for free and the response's latency -- one ladder of 3 071
// squarings modulo p^3 -- becomes one ladder modulo the primes plus one lift of 1 023
inline bool response_by_structure(size_t statements, size_t instances, size_t nb_instances = 0, size_t lt = kChipLanes) {
  return instances >= 4 * statements || (nb_instances != 0 && nb_instances * 16 <= lt);
}
// s = ExtractRandonness on the side stream BESIDE the a^n | x^n launch: only where that launch leaves the second wave slot of
// the SIMDs free (one wave per SIMD or less) or the side launch is a few dozen waves
inline bool extract_beside(size_t nb_statements, size_t nb_instances, size_t lt) {
  return (nb_statements + nb_instances) * 2 <= lt || nb_statements * 2 * 8 <= lt;
}

// A ciphertext range wanted under S shares (threshold.cpp, a ciphertext-major shard): ONE shared chain of squarings carries S window
// products per window on its critical path (4 103 squarings + S x 586 products of 1.5 squarings each for config 4).  Where the chip
// has room -- eight lanes per number and still a SIMD per wave for every group -- the shares split into up to three groups with a
// chain each: 4 096 ciphertexts x 3 shares in two groups (2 + 1), 2 048 in three.
inline int shared_chain_groups(size_t nbs, int S, size_t lt, bool have8) {
  if (!have8 || S < 2 || nbs == 0) return 1;
  const size_t room = lt / (nbs * 8);
  return (int)std::max<size_t>(1, std::min<size_t>(std::min(S, 3), room));
}

// Launches of one call that run BESIDE each other (main stream + side lanes): a launch of at most kExclusiveMaxBlocks workgroups asks
// for the whole LDS of a compute unit per workgroup (run_vm), so that it lands on CUs of its own: the dispatcher otherwise starts every
// queue's workgroups from the same CUs, and three latency-bound launches share the SIMDs of a quarter of the chip while the rest idles.
// Measured on the prover (ms per call, dispatcher's placement | a CU per workgroup): 2 048 instances 74.9 | 58.5, 4 096: 77.5 | 71.1,
// 8 192: 97.4 | 93.1, 12 288: 115.5 | 113.8, 16 384: 130.3 | 129.0, 1 536 x 40: 203 | 200 -- wider launches keep their placement (two
// workgroups per CU beat a second round), so the rule holds for every call size.
constexpr uint32_t kExclusiveMaxBlocks = 128;
// How much of a CU's LDS a workgroup of an assembly launch asks for (run_vm -> launch_vm_asm): 0 the kernel's own, 1 all of it (a CU per
// workgroup), 2 just over half (no two workgroups of the launch on one CU; a side lane's 30 - 60 KB still fit beside it).
//   blocks        workgroups of the launch
//   stream_cus    compute units the launch's stream may use: kChipCUs, or the slice of a context with a CU partition.  A launch that
//                 asks for a CU per workgroup on FEWER CUs than it has workgroups would run in rounds, so that rule stops at stream_cus;
//                 the half-a-CU rule is for contexts that have the whole chip (eight contexts on 32 CUs each
//                 keep the dispatcher's placement)
//   on_side       the launch goes to a side lane of the context
//   in_exclusive_call  a protocol call whose launches run beside each other is in progress (the DDLEQ prover)
//   products      Montgomery products of the program: short programs (links between ladders) are not worth a placement
//   enabled / spread_enabled  the context flags "exclusive" and "spread"
//   exclusive_short   (context flag, default 1) short programs -- the links between ladders -- take a CU per workgroup as well.  A workgroup
//                 that asks for a whole CU waits until one is EMPTY, so with the flag on the side lanes' chains only move between the main
//                 stream's launches (the inversion tree's root reaches the host at 48 ms of a 16 384-instance call, at 38 ms with the flag
//                 off); with it off they run beside the ladders and slow them by as much: Alpha is known 2.4 ms later, the side lane is
//                 done 1.9 ms earlier.  Measured, ms per call, on | off: 16 384 instances 120.6 | 120.6, 8 192: 91.5 | 92.5, 2 048: 54.5 |
//                 55.0, 1 536 x 40: 181.7 | 183.9 -- the chip is the bottleneck either way; round 4's rule stays.
inline int lds_share(uint32_t blocks, uint32_t stream_cus, bool on_side, bool in_exclusive_call, uint64_t products, bool enabled,
                     bool spread_enabled, bool exclusive_short = true) {
  if (!enabled) return 0;
  // (a LINK of the MAIN stream never asks for an empty CU: while a side lane's ladder sits on every CU -- ct1's decryption beside the
  // a^n | x^n ladders -- the main stream's chain between two of its ladders would wait for it: 9 ms between a^n | x^n and X modulo the
  // primes instead of 2, kernel trace of round 5)
  if (in_exclusive_call && blocks <= kExclusiveMaxBlocks && blocks <= stream_cus && (products >= 256 || (exclusive_short && on_side))) return 1;
  // (inside a CU partition the mask already keeps other contexts off these CUs, and the request only delays placement: eight contexts
  // on 32 CUs each, 1 024 / 2 048 ciphertexts per call: 39.4 / 62.1 ms without it, 45.7 / 65.7 ms with it)
  // (the same for a side lane's LADDER inside such a call -- ct1's decryption of 8 192 statements, 256 workgroups: with the dispatcher's
  // placement some CUs get two of them and both wave slots of their SIMDs, and the main stream's links wait for a slot: 25.5 -> 13.3 ms
  // for the launch, 78.6 -> 75.2 ms for the call)
  if (spread_enabled && (!on_side || in_exclusive_call) && stream_cus == kChipCUs && blocks <= stream_cus && products >= 256) return 2;
  return 0;
}

// The late response of the prover (ddleq.cpp respond / struct_response): three latency-bound launches of one wave per SIMD or less --
// the ladder modulo the primes on CUs of its own (lds_share above), b's decryption beside it, the lift behind the ladder while the
// decryption is still running.  The decryption asks for just over half a CU's LDS: never two of its workgroups on a CU, so that EVERY
// CU keeps a wave slot (183 VGPRs) for a workgroup of the lift, which keeps the kernel's own LDS size and fits beside it.  (With the
// dispatcher's placement some CUs get two workgroups of the decryption and a quarter of the lift waits for them: 117 or 124 ms per
// 16 384-instance call, call by call.)
constexpr int kLateDecryptionLds = 2;
constexpr int kLateLiftLds = 0;

// ---- the generic kernels: lanes per number -------------------------------------------------------------------------------------
// The same WT limbs can be sliced over more lanes (WL/2 limbs x 2K lanes).  The natural shape has the cheapest squarings (K == 1:
// triangular rows; the wave-sliced two-slice kernels: every limb product once) and a single wave per SIMD already issues at ~88 % of
// the two-wave rate, so it wins from one wave per SIMD upwards; below that the finer slicing wins (tools/occupancy_sweep.py:
// Decrypt-2048 at 32 768: 1.24 M/s natural vs 1.01 M/s re-sliced; at 16 384: 0.63 vs 0.88 M/s).  Slices stay >= 37 limbs, K <= 4.
// wave_sliced_ok: the 74-limb two-slice shape runs on the wave-sliced assembly kernel (context flags "asm" and "w74"); without it
// those moduli take four lanes of 37 limbs.
struct GenericShape { int WL, K; };
inline GenericShape generic_shape(int WL, int K, size_t launch_nb, size_t segs, size_t lt, bool wave_sliced_ok) {
  while (launch_nb * (size_t)K * segs < lt && K < 4 && WL % 2 == 0 && WL / 2 >= 37) { WL /= 2; K *= 2; }
  if (WL == 74 && K == 2 && !wave_sliced_ok) { WL = 37; K = 4; }
  return {WL, K};
}

}  // namespace plan
