"""The key holder's powers modulo n^3 through the STRUCTURE of the unit group (paillier_amd/csrc/ddleq.cpp struct_pow_n3): the DDLEQ
prover's sanity value ct1^(a^n) b^(n^2) and Alpha = ct1^(x^n) y^(n^2) (ddleq.go:62-87) as (1 + n)^(m e) * omega(X mod n) -- the
plaintext m of ct1 once per statement, ladders modulo the primes with exponents modulo p - 1, q - 1 (number-major window tables on
the generic one-lane kernel), one Teichmueller lift per number.  Must be the reference's integers for every unit input, and must
hand non-units to the literal ladders.  Checked against pow() and against the same call with the path switched off."""
import random

import pytest

from oracle import paillier_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


def _statements(sk_o, count, rng):
    n = sk_o.N
    ct1 = [po.encrypt_with_r_at_level(sk_o, po.encrypt_with_r(sk_o, rng.randrange(n), po.rand_unit(n, rng)).C,
                                      po.rand_unit(n, rng), po.ENC_LEVEL_TWO).C for _ in range(count)]
    a_s = [po.rand_unit(n, rng) for _ in range(count)]
    b_s = [po.rand_unit(n, rng) for _ in range(count)]
    ct2 = [po.nested_randomize_with_ab(sk_o, po.Ciphertext(c, 1), a, b).C for c, a, b in zip(ct1, a_s, b_s)]
    return ct1, ct2, a_s, b_s


def _alpha(n, c1, x, y):
    n2, n3 = n * n, n ** 3
    return pow(c1, pow(x, n, n2), n3) * pow(y, n2, n3) % n3


@pytest.mark.parametrize("bits,S,secpar", [(2048, 10, 1), (2048, 3, 8), (1024, 6, 4), (1024, 9, 2), (3072, 2, 1)])
def test_alpha_by_structure_is_pow(ctx, bits, S, secpar):
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(bits, bits + 31)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(bits + S)
    ct1, ct2, a_s, b_s = _statements(sk_o, S, rng)
    # edge exponents among the draws: x = 1 (x^n = 1: Alpha = ct1 y^(n^2)), x = n - 1 (x^n = n^2 - 1)
    xs = [[po.rand_unit(n, rng) for _ in range(secpar)] for _ in range(S)]
    ys = [[po.rand_unit(n, rng) for _ in range(secpar)] for _ in range(S)]
    xs[0][0], ys[1 % S][0], xs[2 % S][-1] = 1, 1, n - 1
    out = {}
    for struct in (1, 0):
        ctx.set_flag("struct", struct)
        try:
            out[struct] = sk.ProveDDLEQBatch(secpar, ct1, ct2, a_s, b_s, xs, ys)
        finally:
            ctx.set_flag("struct", 1)
    assert out[1] == out[0]
    al, es, fs = out[1]
    for j in range(min(S, 6)):                      # (pow() modulo n^3 is the slow part of this test: a sample, incl. the edge draws)
        assert al[j][0] == _alpha(n, ct1[j], xs[j][0], ys[j][0]), j
    assert al[2 % S][-1] == _alpha(n, ct1[2 % S], xs[2 % S][-1], ys[2 % S][-1])
    # the whole instance -- Alpha, E, F, both challenge bits among them -- against the restatement of proveDDLEQInstance; from four
    # instances per statement the RESPONSE goes through the structure as well (struct_response)
    # The sample: the first instances, plus the first instance of EACH challenge bit (picked by the hash of the transcript, as
    # tests/test_gpu_nonunit.py does; the restatement then recomputes Alpha, the bit and the response on its own).  Every
    # parametrisation above draws both bits (seeds checked with the oracle), and the test insists on it.
    sample = [(j, k) for j in range(min(S, 2)) for k in range(min(secpar, 4))]
    for want in (True, False):
        hit = next(((j, k) for j in range(S) for k in range(secpar)
                    if po.random_oracle_bit(ct1[j], ct2[j], xs[j][k], ys[j][k], al[j][k]) == want), None)
        assert hit is not None, f"no instance with challenge bit {int(want)} among the draws: change the seed"
        if hit not in sample:
            sample.append(hit)
    bits_seen = set()
    for j, k in sample:
        inst = po.prove_ddleq_instance_xy(sk_o, po.Ciphertext(ct1[j], 1), po.Ciphertext(ct2[j], 1), a_s[j], b_s[j], xs[j][k], ys[j][k])
        assert (al[j][k], es[j][k], fs[j][k]) == (inst.Alpha, inst.E, inst.F), (j, k)
        bits_seen.add(inst.E != xs[j][k])
    assert bits_seen == {True, False}


def test_non_units_fall_back_to_the_literal_ladders(ctx):
    """y a multiple of q, ct1 a multiple of p (not a unit: no plaintext, the structure theorem does not apply): the flags of the
    exact divisions send the call down the ladders on ct1 itself -- the reference's integers, whatever the inputs."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(2048, 2079)
    n, n2, n3 = sk_o.N, sk_o.N ** 2, sk_o.N ** 3
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(5)
    S = 6
    ct1, ct2, a_s, b_s = _statements(sk_o, S, rng)
    xs, ys = [po.rand_unit(n, rng) for _ in range(S)], [po.rand_unit(n, rng) for _ in range(S)]
    # a non-unit draw y whose challenge bit is 0 (the prover then needs Alpha and the sanity check only, no response): redrawn
    # until the hash says so, so that lane 3 is ALWAYS among the instances proved
    for _ in range(64):
        ys[3] = q * rng.randrange(1, p)
        if not po.random_oracle_bit(ct1[3], ct2[3], xs[3], ys[3], _alpha(n, ct1[3], xs[3], ys[3])):
            break
    keep = [i for i in range(S) if not po.random_oracle_bit(ct1[i], ct2[i], xs[i], ys[i], _alpha(n, ct1[i], xs[i], ys[i]))]
    assert 3 in keep and len(keep) >= 2
    pick = lambda v: [v[i] for i in keep]
    al, es, fs = sk.ProveDDLEQInstancesBatch(pick(ct1), pick(ct2), pick(a_s), pick(b_s), pick(xs), pick(ys))
    assert al == [_alpha(n, ct1[i], xs[i], ys[i]) for i in keep] and es == pick(xs) and fs == pick(ys)
    # a statement whose ct1 is not a unit, with a ct2 that IS ct1^(a^n) b^(n^2): the sanity check passes on the literal path
    c1 = p * rng.randrange(1, n2)
    a, b = po.rand_unit(n, rng), po.rand_unit(n, rng)
    c2 = pow(c1, pow(a, n, n2), n3) * pow(b, n2, n3) % n3
    for _ in range(40):
        x, y = po.rand_unit(n, rng), po.rand_unit(n, rng)
        if not po.random_oracle_bit(c1, c2, x, y, _alpha(n, c1, x, y)):
            break
    al, es, fs = sk.ProveDDLEQInstancesBatch([ct1[0], c1], [ct2[0], c2], [a_s[0], a], [b_s[0], b], [xs[0] if 0 in keep else x, x],
                                             [ys[0] if 0 in keep else y, y])
    assert al[1] == _alpha(n, c1, x, y)
    # and a false statement is still refused (ddleq.go:68)
    with pytest.raises(pa.PaillierHipError, match="cannot prove re-encryption"):
        sk.ProveDDLEQInstancesBatch([ct1[0]], [ct2[1]], [a_s[0]], [b_s[0]], [xs[0]], [ys[0]])


def test_generic_one_lane_kernel_number_major_windows(ctx):
    """The ladders modulo the primes run per-number 4-bit windows on NUMBER-major tables (VM_STORET / VM_MULVT) of the generic
    one-lane kernel vm_asm_37_1 ("prime_lanes" 0: a batch this small would take the four-lane twins, which the next test covers);
    nm4 = 0 puts them back on limb-major tables: the same proofs."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(2048, 2080)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(6)
    S = 300
    ct1, ct2, a_s, b_s = _statements(sk_o, 4, rng)
    rep = lambda v: [v[i % 4] for i in range(S)]
    xs, ys = [po.rand_unit(n, rng) for _ in range(S)], [po.rand_unit(n, rng) for _ in range(S)]
    got = {}
    ctx.set_flag("prime_lanes", 0)
    try:
        for nm4 in (1, 0):
            ctx.set_flag("nm4", nm4)
            got[nm4] = sk.ProveDDLEQInstancesBatch(rep(ct1), rep(ct2), rep(a_s), rep(b_s), xs, ys)
            assert ctx.last_vm_asm() == ctx.last_vm_launches()
    finally:
        ctx.set_flag("nm4", 1)
        ctx.set_flag("prime_lanes", 1)
    assert got[1] == got[0]
    assert got[1][0][:8] == [_alpha(n, rep(ct1)[i], xs[i], ys[i]) for i in range(8)]
    assert all(pk.VerifyDDLEQInstancesBatch(rep(ct1), rep(ct2), xs, ys, *got[1]))


@pytest.mark.parametrize("bits,S,secpar", [(1024, 7, 1), (1024, 5, 2), (2048, 4, 3)])
def test_late_response_through_the_structure(ctx, bits, S, secpar):
    """Round 5: with fewer than four instances per statement and a batch that fills the chip the response goes through the structure of
    the unit group AFTER the hash -- b's plaintext for the statements that have an instance with challenge bit 1 only (flag "late").
    Small batches take the early form, so the occupancy target is set to one lane ("lanes_wanted" 1: every batch 'fills the chip') to
    reach the late form here; the proofs must equal the one-ladder response's (late 0), the restatement of proveDDLEQInstance for EVERY
    instance, and verify.  secpar 2 and 3: several instances share a statement (its s, b gathered once)."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(bits, bits + 77)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(bits + 7 * S + secpar)
    ct1, ct2, a_s, b_s = _statements(sk_o, S, rng)
    xs = [[po.rand_unit(n, rng) for _ in range(secpar)] for _ in range(S)]
    ys = [[po.rand_unit(n, rng) for _ in range(secpar)] for _ in range(S)]
    out, launches = {}, {}
    ctx.set_flag("lanes_wanted", 1)
    try:
        for late in (1, 0):
            ctx.set_flag("late", late)
            out[late] = sk.ProveDDLEQBatch(secpar, ct1, ct2, a_s, b_s, xs, ys)
            launches[late] = ctx.last_vm_launches()
    finally:
        ctx.set_flag("late", 1)
        ctx.set_flag("lanes_wanted", 0)
    assert out[1] == out[0]
    assert launches[1] != launches[0], "the flag did not change the path: the late form was not reached"
    al, es, fs = out[1]
    bits_seen = set()
    for j in range(S):
        for k in range(secpar):
            inst = po.prove_ddleq_instance_xy(sk_o, po.Ciphertext(ct1[j], 1), po.Ciphertext(ct2[j], 1), a_s[j], b_s[j], xs[j][k], ys[j][k])
            assert (al[j][k], es[j][k], fs[j][k]) == (inst.Alpha, inst.E, inst.F), (j, k)
            bits_seen.add(inst.E != xs[j][k])
    assert bits_seen == {True, False}, "the draws must hit both challenge bits: change the seed"
    flat = lambda v: [x for row in v for x in row]
    rep = lambda v: [v[j] for j in range(S) for _ in range(secpar)]
    assert all(pk.VerifyDDLEQInstancesBatch(rep(ct1), rep(ct2), flat(xs), flat(ys), flat(al), flat(es), flat(fs)))


@pytest.mark.parametrize("bits", [2048, 1024])
def test_ladders_modulo_the_primes_on_four_lanes(ctx, bits):
    """Round 5: a batch that leaves most SIMDs empty at one lane per number runs the key holder's ladders modulo the primes (37 limbs)
    on their four-lane twins -- p, q as 40-limb moduli in four lanes of 10, vm_asm_10_4 / vm_kernel<10, 4> (plan::prime_lanes, flag
    "prime_lanes") --: x^n through the primes (key holder's Encrypt, both levels), the prover's X modulo the primes with per-number windows
    on number-major AND limb-major tables, its response, s = ExtractRandonness.  Same integers as one lane per number (flag 0), as the
    compiler-generated kernel (asm 0), as the oracle."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(bits, bits + 91)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(bits + 4)
    S, secpar = 5, 2
    ct1, ct2, a_s, b_s = _statements(sk_o, S, rng)
    xs = [[po.rand_unit(n, rng) for _ in range(secpar)] for _ in range(S)]
    ys = [[po.rand_unit(n, rng) for _ in range(secpar)] for _ in range(S)]
    ms = [rng.randrange(n) for _ in range(70)] + [0, n - 1]
    rs = [po.rand_unit(n, rng) for _ in ms]
    rs[3], rs[4] = 1, n - 1
    ms2 = [rng.randrange(n * n) for _ in range(9)]
    got, mads = {}, {}
    try:
        for lanes, asm, nm4 in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0)):
            ctx.set_flag("prime_lanes", lanes)
            ctx.set_flag("asm", asm)
            ctx.set_flag("nm4", nm4)
            e1 = sk.EncryptWithRBatch(ms, rs)
            mads[(lanes, asm, nm4)] = ctx.last_profile()["vm_mads"]
            if asm:
                assert ctx.last_vm_asm() == ctx.last_vm_launches()
            e2 = sk.EncryptWithRBatch(ms2, rs[:len(ms2)], po.ENC_LEVEL_TWO)
            pr = sk.ProveDDLEQBatch(secpar, ct1, ct2, a_s, b_s, xs, ys)
            mads[(lanes, asm, nm4)] += ctx.last_profile()["vm_mads"]          # (a 1024-bit key's Encrypt has no ladder modulo the primes)
            if asm:
                assert ctx.last_vm_asm() == ctx.last_vm_launches()
            got[(lanes, asm, nm4)] = (e1, e2, pr)
    finally:
        ctx.set_flag("prime_lanes", 1)
        ctx.set_flag("asm", 1)
        ctx.set_flag("nm4", 1)
    ref = got[(1, 1, 1)]
    assert all(v == ref for v in got.values())
    assert mads[(1, 1, 1)] != mads[(0, 1, 1)], "the flag did not change the shape of the ladders modulo the primes"
    assert ref[0] == [po.encrypt_with_r(sk_o, m, r).C for m, r in zip(ms, rs)]
    assert ref[1] == [po.encrypt_with_r_at_level(sk_o, m, r, po.ENC_LEVEL_TWO).C for m, r in zip(ms2, rs)]
    al, es, fs = ref[2]
    bits_seen = set()
    for j in range(S):
        for k in range(secpar):
            inst = po.prove_ddleq_instance_xy(sk_o, po.Ciphertext(ct1[j], 1), po.Ciphertext(ct2[j], 1), a_s[j], b_s[j], xs[j][k], ys[j][k])
            assert (al[j][k], es[j][k], fs[j][k]) == (inst.Alpha, inst.E, inst.F), (j, k)
            bits_seen.add(inst.E != xs[j][k])
    assert bits_seen == {True, False}
