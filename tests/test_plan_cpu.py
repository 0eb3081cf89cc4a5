"""The planning predicates (paillier_amd/csrc/plan.hpp) on the CPU, through pgpu_plan_query of include/paillier_hip_debug.h.

Every size decision of the engine -- lanes per number, window widths, whether a ladder is split -- is a pure function there, and
the protocol bodies call the same functions.  Round 3 shipped a regression in exactly such a predicate (the split gate of the
prover's Alpha ladder bounded the 192 slots a 7-bit table occupies where the window-width choice bounded the 128 entries its
gathers address: between 50 121 and 74 986 numbers the p-adic split was dropped, secpar-40 prove 204 k -> 166 k instances/s);
these tests pin the decisions at the batch sizes BASELINE.json's configs produce and at the boundaries in between.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H37, WT3_P = 37, 110        # 2048-bit key: limbs of a prime, limbs of p^3 (two slices of 55)
H74 = 74                    # limbs of n


@pytest.fixture(scope="module")
def q():
    import __graft_entry__ as ge
    ge.build()
    from paillier_amd import api
    return api.plan_query


def pad(n):
    return (n + 255) // 256 * 256


# numbers per launch of the prover's Alpha ladder (statements + instances, both padded to 256):
#   secpar 1, 16 384 statements: 32 768;  secpar 40, 1 536 statements: 1 536 + 61 440 = 62 976 (the bench shape);
#   rank of 8 at secpar 1: 4 096;  tools/prove_only.py 65 536 / 98 304: 131 072 / 196 608
ALPHA = [
    # nb        triple win nm5 split
    (4096,      1,     7,  0,  0),      # below one wave per SIMD for the p^2 stage: one ladder is shorter than two
    (8192,      1,     7,  0,  0),
    (16384,     1,     7,  0,  1),      # 4 nb = 65 536 lanes: the split starts here
    (32768,     1,     7,  0,  1),
    (pad(50000), 1,    7,  0,  1),
    (50176,     1,     7,  0,  1),      # the first size the round-3 regression dropped (x193 bound: 50 120)
    (62976,     1,     7,  0,  1),      # BENCH ddleq_prove_2048_secpar40
    (pad(74000), 1,    7,  0,  1),
    (74752,     1,     7,  0,  1),      # last multiple of 256 with 7-bit tables inside the gather offsets (74 986)
    (75008,     1,     5,  1,  1),      # beyond: 5-bit windows on number-major tables, the split STAYS
    (80000,     1,     5,  1,  1),
    (98304,     1,     5,  1,  1),
    (131072,    1,     5,  1,  1),
    (196608,    1,     5,  1,  1),
    (262144,    1,     5,  1,  1),
    (524288,    0,     5,  1,  0),      # the digit kernel's own limb-major gathers no longer fit: generic kernels
]


@pytest.mark.parametrize("nb,triple,win,nm5,split", ALPHA)
def test_alpha_ladder_plan(q, nb, triple, win, nm5, split):
    assert q("crt3_ladder", nb, H37, WT3_P, 0, 1, 1) == [triple, win, nm5, split]


def test_split_gate_and_window_choice_agree_everywhere(q):
    """The two predicates that disagreed in round 3: wherever the window choice says 7 bits the split gate must accept the 7-bit
    table, and wherever it says 5 the 5-bit one -- for every multiple of 256 up to 300 000 numbers and both digit widths."""
    for H, wt3 in ((H37, WT3_P), (H74, 222)):
        for nb in range(16384, 300001, 256):
            triple, win, nm5, split = q("crt3_ladder", nb, H, wt3, 0, 1, 1)
            assert win == q("triple_window_bits", nb, H)[0]
            assert nm5 == (win == 5)
            if triple:
                assert split == 1, (H, nb, win)


def test_gathers_address_entries_not_slots(q):
    """A 7-bit table OCCUPIES 128 + 64 slots; its gathers ADDRESS 128 entries (+ the lane's own offset): only those bound nb."""
    assert q("perlane_table_slots", 7, 0) == [192] and q("gather_entries", 7) == [129]
    assert q("perlane_table_slots", 5, 1) == [48] and q("gather_entries", 5) == [33]
    assert q("perlane_table_slots", 4, 1) == [17] and q("perlane_table_slots", 4, 0) == [16] and q("perlane_table_slots", 5, 0) == [32]
    # 7-bit windows on the digit kernels: up to 74 987 numbers for 37-limb digits, 37 493 for 74-limb digits
    assert q("triple_window_bits", 74987, H37) == [7] and q("triple_window_bits", 74988, H37) == [5]
    assert q("triple_window_bits", 37493, H74) == [7] and q("triple_window_bits", 37494, H74) == [5]


def test_shared_exponent_ladders_without_per_number_windows(q):
    for nb in (8192, 65536, 131072):
        triple, win, nm5, split = q("crt3_ladder", nb, H37, WT3_P, 0, 0, 1)
        assert (win, nm5, split) == (5, 0, 0) and triple == 1


def test_split_needs_its_prerequisites_and_the_occupancy_target(q):
    assert q("crt3_ladder", 62976, H37, WT3_P, 0, 1, 0)[3] == 0            # lift off / no pair kernel for p^2 ...
    assert q("crt3_ladder", 4096, H37, WT3_P, 1, 1, 1)[3] == 1             # lanes_wanted = 1 (tests force the split on toy batches)
    assert q("crt3_ladder", 16128, H37, WT3_P, 0, 1, 1)[3] == 0            # just below one wave per SIMD


RESPONSE = [
    # nb (bit-1 instances)  usable split
    (256,    1, 0),
    (8192,   1, 0),     # secpar 1 x 16 384: about half the instances
    (8448,   1, 0),
    (16384,  1, 1),
    (30720,  1, 1),     # secpar 40 x 1 536: about half of 61 440
    (61440,  1, 1),
    (74752,  1, 1),
    (75008,  0, 0),     # two per-number tables of 7-bit windows: beyond the offsets the literal sequence takes over
]


@pytest.mark.parametrize("nb,usable,split", RESPONSE)
def test_response_ladder_plan(q, nb, usable, split):
    assert q("crt3_two", nb, H37, 0, 1) == [usable, split]


def test_early_response_only_where_its_kernels_are_certain(q):
    assert q("early_response_ok", 16384, H37) == [1] and q("early_response_ok", 61440, H37) == [1]
    assert q("early_response_ok", 98304, H37) == [0]


def test_response_through_the_structure_of_the_unit_group(q):
    """From four instances per statement (the per-statement decryptions of s and b amortise), and for batches so small that the
    chip is mostly idle (the response's latency: one ladder modulo the primes + one lift instead of 3 071 squarings modulo p^3)."""
    assert q("response_by_structure", 1536, 61440, 61440, 0) == [1]          # BENCH ddleq_prove_2048_secpar40
    assert q("response_by_structure", 16384, 16384, 16384, 0) == [0]         # BENCH ddleq_prove_2048 (secpar 1)
    assert q("response_by_structure", 4, 16, 256, 0) == [1] and q("response_by_structure", 5, 15, 256, 0) == [1]
    assert q("response_by_structure", 2048, 2048, 2048, 0) == [1]            # a rank's 2 048 instances at N = 8
    assert q("response_by_structure", 4096, 4096, 4096, 0) == [1] and q("response_by_structure", 4352, 4352, 4352, 0) == [0]
    assert q("response_by_structure", 8192, 8192, 8192, 0) == [0]
    assert q("response_by_structure", 16384, 49152, 49152, 0) == [0] and q("response_by_structure", 16384, 65536, 65536, 0) == [1]


def test_extract_randomness_beside_the_first_launch(q):
    assert q("extract_beside", 16384, 16384, 0) == [1]       # secpar 1 x 16 384: the a^n | x^n launch is one wave per SIMD
    assert q("extract_beside", 1536, 61440, 0) == [1]        # secpar 40: the side launch is a few dozen waves
    assert q("extract_beside", 32768, 32768, 0) == [0]       # both wave slots taken: s follows on the main stream
    assert q("extract_beside", 2048, 2048, 0) == [1]
    # the verifier's x^(e0) W^n modulo n^3 as two eight-lane ladders side by side: up to 4 096 numbers
    assert q("dual_n3_two_ladders", 2048, 0) == [1]
    assert q("dual_n3_two_ladders", 4096, 0) == [1]
    assert q("dual_n3_two_ladders", 8192, 0) == [0]
    # a ciphertext-major threshold shard: groups of shares with a chain each while every group still has a SIMD per wave at 8 lanes
    assert q("shared_chain_groups", 16384, 3, 0, 1) == [1]
    assert q("shared_chain_groups", 8192, 3, 0, 1) == [1]
    assert q("shared_chain_groups", 4096, 3, 0, 1) == [2]
    assert q("shared_chain_groups", 2048, 3, 0, 1) == [3]
    assert q("shared_chain_groups", 256, 2, 0, 1) == [2]
    assert q("shared_chain_groups", 2048, 3, 0, 0) == [1]


def test_placement_by_lds_size(q):
    """plan::lds_share(blocks, stream_cus, on_side, in_exclusive_call, products, exclusive_flag, spread_flag)."""
    # inside a prover call: a compute unit per workgroup for launches of up to 128 workgroups, main stream or side lane -- links
    # between ladders (a few products) included unless the flag "exclusive_short" is off
    assert q("lds_share", 64, 256, 1, 1, 12, 1, 1) == [1]
    assert q("lds_share", 64, 256, 1, 1, 12, 1, 1, 0) == [0]
    assert q("lds_share", 64, 256, 0, 1, 12, 1, 1) == [2] or q("lds_share", 64, 256, 0, 1, 12, 1, 1) == [0]   # a link of the MAIN stream: never a whole CU
    assert q("lds_share", 64, 256, 0, 1, 12, 1, 1) != [1]
    assert q("lds_share", 64, 256, 1, 1, 4000, 1, 1) == [1]
    assert q("lds_share", 128, 256, 0, 1, 4000, 1, 1) == [1]
    # wider: a ladder of at most one workgroup per CU spreads (just over half a CU's LDS) -- the main stream's, and inside a prover call a
    # side lane's as well (ct1's decryption of 8 192 statements: 256 workgroups; two on a CU take both wave slots of its SIMDs and the
    # main stream's links wait for one); outside such a call a side lane's launch keeps its size, and so does every link
    assert q("lds_share", 256, 256, 0, 1, 4000, 1, 1) == [2]
    assert q("lds_share", 256, 256, 1, 1, 4000, 1, 1) == [2]
    assert q("lds_share", 256, 256, 1, 0, 4000, 1, 1) == [0]
    assert q("lds_share", 256, 256, 1, 1, 12, 1, 1) == [0]
    assert q("lds_share", 512, 256, 1, 1, 4000, 1, 1) == [0]
    assert q("lds_share", 257, 256, 0, 0, 4000, 1, 1) == [0]
    # outside a prover call: the spread rule alone, for ladders (>= 256 products), never for the links between them
    assert q("lds_share", 64, 256, 0, 0, 4000, 1, 1) == [2]
    assert q("lds_share", 64, 256, 0, 0, 100, 1, 1) == [0]
    # the switches
    assert q("lds_share", 64, 256, 0, 1, 4000, 0, 1) == [0]
    assert q("lds_share", 64, 256, 0, 0, 4000, 1, 0) == [0]
    assert q("lds_share", 64, 256, 0, 1, 4000, 1, 0) == [1]
    # a context confined to 32 CUs (eight contexts side by side) must not ask
    # for a CU -- or half of one -- per workgroup with 64 workgroups: that would be two rounds on its slice
    assert q("lds_share", 64, 32, 0, 0, 4000, 1, 1) == [0]
    assert q("lds_share", 64, 32, 0, 1, 4000, 1, 1) == [0]
    assert q("lds_share", 32, 32, 0, 0, 4000, 1, 1) == [0]           # measured: 62.1 ms without the request, 65.7 with it
    assert q("lds_share", 32, 32, 0, 1, 4000, 1, 1) == [1]


def test_lanes_target_follows_the_cu_partition(q):
    """A context confined to a slice of the compute units plans for that slice."""
    assert q("lanes_target", 0, 256) == [65536]
    assert q("lanes_target", 0, 32) == [8192]
    assert q("lanes_target", 1, 32) == [1]                       # an explicit "lanes_wanted" wins
    # eight contexts on 32 CUs each, 2 048 ciphertexts per PartialDecrypt call: four lanes per number (one wave per SIMD of the slice)
    assert q("pair_lanes_shared", 2048, 8192, 1, 1) == [4]
    assert q("pair_lanes_shared", 2048, 0, 1, 1) == [8]          # the same call with the whole chip to itself


def test_generic_kernel_lanes(q):
    """plan::generic_shape: natural shape from one wave per SIMD upwards, finer slices (>= 37 limbs, <= 4 lanes) below."""
    assert q("generic_shape", 148, 1, 65536, 1, 0, 1) == [148, 1]
    assert q("generic_shape", 148, 1, 32768, 1, 0, 1) == [74, 2]
    assert q("generic_shape", 148, 1, 16384, 1, 0, 1) == [37, 4]
    assert q("generic_shape", 148, 1, 16384, 2, 0, 1) == [74, 2]          # two segments share the launch
    assert q("generic_shape", 74, 1, 16384, 2, 0, 1) == [37, 2]
    assert q("generic_shape", 37, 1, 256, 1, 0, 1) == [37, 1]             # 37-limb slices are the narrowest
    assert q("generic_shape", 110, 1, 256, 1, 0, 1) == [55, 2]
    assert q("generic_shape", 148, 1, 256, 1, 1, 1) == [148, 1]           # lanes_wanted = 1: always the natural shape
    assert q("generic_shape", 74, 2, 65536, 1, 0, 0) == [37, 4]           # without the wave-sliced kernel: four lanes of 37
    assert q("generic_shape", 74, 2, 65536, 1, 0, 1) == [74, 2]


PAIR_SHARED = [
    # numbers  have4 have8 -> lanes
    (65536, 1, 1, 2), (32768, 1, 1, 2), (32512, 1, 1, 4), (16384, 1, 1, 4), (12288, 1, 1, 4), (8448, 1, 1, 4),
    (8192, 1, 1, 8), (6144, 1, 1, 8), (2048, 1, 1, 8), (256, 1, 1, 8),
    (8192, 1, 0, 4), (2048, 0, 0, 2), (16384, 0, 1, 2),
]


@pytest.mark.parametrize("numbers,have4,have8,lanes", PAIR_SHARED)
def test_pair_lanes_for_shared_exponents(q, numbers, have4, have8, lanes):
    """PartialDecrypt / Encrypt modulo n^2: two lanes from one wave per SIMD upwards, four below, eight while every wave still
    has a SIMD of its own (a rank's 6 144 units of BASELINE config 4 at N = 8)."""
    assert q("pair_lanes_shared", numbers, 0, have4, have8) == [lanes]
    if not have8:
        assert q("pair_lanes_2or4", numbers, 0, have4) == [lanes]
    # sixteen lanes (round 5) only where TWO such ladders side by side still leave every wave a SIMD of its own: up to 2 048 numbers, and
    # only on top of eight
    want16 = 16 if (have4 and have8 and numbers * 32 <= 65536) else lanes
    assert q("pair_lanes_shared", numbers, 0, have4, have8, 1) == [want16]


def test_pair_kernel_serves(q):
    assert q("pair_kernel_serves", 32768, 0, 0) == [1] and q("pair_kernel_serves", 16384, 0, 0) == [0]
    assert q("pair_kernel_serves", 256, 0, 1) == [1]


CRT_LANES = [
    # nb      lanes usable      (one-lane kernel of a 2048-bit key with its two-lane variant)
    (65536, 1, 1), (32768, 1, 1), (30720, 1, 1), (20480, 1, 1), (16640, 1, 1), (16384, 2, 1), (8192, 2, 1), (256, 2, 1),
]


@pytest.mark.parametrize("nb,lanes,usable", CRT_LANES)
def test_crt_halves_lane_choice(q, nb, lanes, usable):
    """Decrypt-2048 (the headline at 65 536): one lane per number down to 16 385 ciphertexts, two below."""
    assert q("crt_pair_lanes", 1, 1, nb, 0) == [lanes, usable]


def test_ladders_modulo_the_primes_lane_choice(q):
    """The key holder's ladders modulo the 37-limb primes (both halves in one launch): four lanes of 10 limbs per number while that
    leaves every wave a SIMD of its own -- up to 8 192 numbers on the whole chip (the prover's response at 16 384 instances, every
    ladder modulo the primes of a rank's 2 048 instances at N = 8) --, one lane on the unrolled kernel above; never without the twins
    (other key sizes) or with the flag off; and a context confined to a CU partition counts its own compute units."""
    assert q("prime_lanes", 8192, 0, 1, 1) == [4] and q("prime_lanes", 8448, 0, 1, 1) == [1]
    assert q("prime_lanes", 256, 0, 1, 1) == [4] and q("prime_lanes", 32768, 0, 1, 1) == [1]
    assert q("prime_lanes", 2048, 0, 0, 1) == [1] and q("prime_lanes", 2048, 0, 1, 0) == [1]
    assert q("prime_lanes", 1024, 32 * 256, 1, 1) == [4] and q("prime_lanes", 2048, 32 * 256, 1, 1) == [1]
    assert q("prime_lanes", 2048, 1, 1, 1) == [1]          # "lanes_wanted" 1: every batch 'fills the chip'
    # beside other launches of the call (X modulo the primes next to ct1's decryption: 2; s on a side lane: 4) the bound shrinks
    assert q("prime_lanes", 4096, 0, 1, 1, 2) == [4] and q("prime_lanes", 4352, 0, 1, 1, 2) == [1]
    assert q("prime_lanes", 2048, 0, 1, 1, 4) == [4] and q("prime_lanes", 2304, 0, 1, 1, 4) == [1]


def test_crt_halves_on_eight_lanes(q):
    """Both CRT halves of up to 8 192 numbers fit two waves per SIMD at eight lanes per number: the ladders of Decrypt and of the key
    holder's x^n take vm_asm_10_96 there (a rank's 2 048 prover instances: 4 096 numbers); never without the constants or the flag."""
    assert q("crt_pair_lanes8", 8192, 0, 1, 1) == [1] and q("crt_pair_lanes8", 8448, 0, 1, 1) == [0]
    assert q("crt_pair_lanes8", 256, 0, 1, 1) == [1]
    assert q("crt_pair_lanes8", 256, 0, 0, 1) == [0] and q("crt_pair_lanes8", 256, 0, 1, 0) == [0]
    assert q("crt_pair_lanes8", 1024, 32 * 256, 1, 1) == [1] and q("crt_pair_lanes8", 1280, 32 * 256, 1, 1) == [0]
    # inside a prover call (other launches beside it): one wave per SIMD at most
    assert q("crt_pair_lanes8", 4096, 0, 1, 1, 2) == [1] and q("crt_pair_lanes8", 4352, 0, 1, 1, 2) == [0]


def test_cube_ladders_on_two_lanes_per_digit(q):
    """The ladders modulo p^3, q^3 of up to 4 096 numbers (both halves: a SIMD per wave at eight lanes per number) take two lanes per digit;
    on a side lane of a prover call -- ct1's decryption beside the main stream's ladders -- up to 2 048."""
    assert q("crt_triple_lanes6", 4096, 0, 1, 1) == [1] and q("crt_triple_lanes6", 4352, 0, 1, 1) == [0]
    assert q("crt_triple_lanes6", 2048, 0, 1, 1, 2) == [1] and q("crt_triple_lanes6", 2304, 0, 1, 1, 2) == [0]
    assert q("crt_triple_lanes6", 256, 0, 0, 1) == [0] and q("crt_triple_lanes6", 256, 0, 1, 0) == [0]


def test_three_digit_kernel_on_four_lanes_per_digit(q):
    """The verifier's two ladders modulo n^3 side by side at sixteen lanes per number each: up to 2 048 numbers."""
    assert q("triple_four_lanes_per_digit", 2048, 0, 1, 1) == [1] and q("triple_four_lanes_per_digit", 2304, 0, 1, 1) == [0]
    assert q("triple_four_lanes_per_digit", 256, 0, 0, 1) == [0] and q("triple_four_lanes_per_digit", 256, 0, 1, 0) == [0]
    # a lone ladder (level-two Encrypt's (r^n)^n): up to 4 096
    assert q("triple_four_lanes_per_digit", 4096, 0, 1, 1, 1) == [1] and q("triple_four_lanes_per_digit", 4352, 0, 1, 1, 1) == [0]


def test_dual_ladder_windows_and_tables(q):
    W2 = 148
    assert q("dual_pair_window_bits", 16384, W2, 1) == [5] and q("dual_pair_window_bits", 61440, W2, 1) == [5]
    assert q("dual_pair_window_bits", 148224, W2, 1) == [4]          # 49 slots of 5-bit windows no longer fit, 18 of 4-bit do
    assert q("dual_pair_window_bits", 402944, W2, 1) == [4] and q("dual_pair_window_bits", 403200, W2, 1) == [0]
    assert q("pair_nm4_fits", 402944, W2) == [1] and q("pair_nm4_fits", 403200, W2) == [0]
    assert q("shared_chain_pays", 8192, 0) == [1] and q("shared_chain_pays", 7936, 0) == [0]
    assert q("triple_two_lanes_per_digit", 8192, 0) == [1] and q("triple_two_lanes_per_digit", 8448, 0) == [0]


def test_unknown_decision_is_an_error(q):
    from paillier_amd.api import PaillierHipError
    with pytest.raises(PaillierHipError):
        q("no_such_decision", 1, 2)
    with pytest.raises(PaillierHipError):
        q("crt3_ladder", 1, 2)          # too few arguments


def test_no_size_predicate_outside_plan_hpp():
    """The protocol bodies must not grow private copies of these bounds again: the 32-bit gather span and the chip's lane count
    appear in plan.hpp only."""
    csrc = os.path.join(ROOT, "paillier_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".cpp", ".hpp")) or f in ("plan.hpp", "hostbig.hpp"):      # (hostbig: the digit base of its long division)
            continue
        src = open(os.path.join(csrc, f)).read()
        src = re.sub(r"//[^\n]*", "", src)
        assert "1ull << 32" not in src and "1024 * 64" not in src, f"{f} carries a size predicate of its own: move it to plan.hpp"
        # the lane-halving loop of the generic kernels and the placement rules live in plan.hpp too
        assert not re.search(r"\bK\s*\*=\s*2\b", src) and "kExclusiveMaxBlocks" not in src, f"{f}: lanes / placement decided outside plan.hpp"


def test_no_environment_switch_changes_the_plan():
    """every switch is a documented pgpu_ctx_set_flag name; the library reads the environment only for the two
    diagnostics that print timings (a Go host inherits its environment from wherever it runs)."""
    csrc = os.path.join(ROOT, "paillier_amd", "csrc")
    allowed = {"PGPU_PROFILE_DUMP", "PGPU_HOST_TRACE"}
    header = open(os.path.join(ROOT, "include", "paillier_hip.h")).read()
    dbg = open(os.path.join(ROOT, "include", "paillier_hip_debug.h")).read()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".cpp", ".hpp", ".hip", ".h")):
            continue
        src = open(os.path.join(csrc, f)).read()
        for name in re.findall(r'getenv\(\s*"([A-Z_0-9]+)"', src):
            assert name in allowed, f"{f} reads {name}: make it a pgpu_ctx_set_flag name"
        assert len(re.findall(r"getenv\(", src)) == len(re.findall(r'getenv\(\s*"', src)), f"{f}: getenv of a computed name"
    for name in allowed:
        assert name in dbg, f"{name} is not documented in paillier_hip_debug.h"
    # every flag name the library accepts is documented in the header
    ctx_src = open(os.path.join(csrc, "ctx.cpp")).read()
    for name in re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', ctx_src):
        assert f'"{name}"' in header, f"flag {name} is not documented in include/paillier_hip.h"
