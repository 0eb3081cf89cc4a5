"""GPU parity for the generalized (Damgard-Jurik s = 2) scheme: level-two EncryptWithR / Decrypt and nested
encryption (paillier.go:199-203,206-218,292-372), against the oracle.  n^3 of a 1024-bit key runs on the 2-lane
kernel shape, n^3 of a 2048-bit key on the 4-lane shape."""
import json
import os
import random

import pytest

from oracle import paillier_oracle as po

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


@pytest.mark.parametrize("bits", ["1024", "2048", "3072"])
def test_level_two_encrypt_decrypt(ctx, bits):
    import paillier_amd as pa
    from paillier_amd import ENC_LEVEL_TWO
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"][bits]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    n2, n3 = n * n, n ** 3
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    rng = random.Random(int(bits) + 2)
    ms = [0, 1, 2, n - 1, n, n + 1, n2 - 1] + [rng.randrange(n2) for _ in range(9)]
    rs = [po.rand_unit(n, rng) for _ in ms]
    cts = pk.EncryptWithRBatch(ms, rs, level=ENC_LEVEL_TWO)
    assert cts == [po.encrypt_with_r_at_level(sk_o, m, r, po.ENC_LEVEL_TWO).C for m, r in zip(ms, rs)]
    assert sk.DecryptBatch(cts, level=ENC_LEVEL_TWO) == ms
    # arbitrary elements of Z_{n^3}: must equal the reference's recovery algorithm on the same input
    weird = [rng.randrange(n3) for _ in range(4)] + [n3 - 1, 1, p, n, n2, 0, 5 * n2]
    got, st = sk.DecryptBatch(weird, level=ENC_LEVEL_TWO, return_status=True)
    want = [po.decrypt(sk_o, po.Ciphertext(c, po.ENC_LEVEL_TWO)) for c in weird]
    assert got == want
    # the default path is CRT over p^3, q^3 with the non-units redone by the reference formula; both must agree with it
    from math import gcd
    assert [bool(s & pa.LANE_NONUNIT) for s in st] == [gcd(c, n) != 1 for c in weird]
    assert sk.DecryptBatch(weird, level=ENC_LEVEL_TWO, flags=pa.DECRYPT_NO_CRT) == want
    many = [rng.randrange(n3) for _ in range(300 if int(bits) <= 2048 else 40)]
    assert sk.DecryptBatch(many, level=ENC_LEVEL_TWO) == sk.DecryptBatch(many, level=ENC_LEVEL_TWO, flags=pa.DECRYPT_NO_CRT)


def test_nested_encrypt_decrypt(ctx):
    """paillier_test.go:65-87: NestedEncrypt = level-one encryption encrypted again at level two."""
    import paillier_amd as pa
    from paillier_amd import ENC_LEVEL_ONE, ENC_LEVEL_TWO
    sk_o, p, q = po.keygen_seeded(1024, 1)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(77)
    ms = [rng.randrange(n) for _ in range(20)]
    inner = pk.EncryptWithRBatch(ms, [po.rand_unit(n, rng) for _ in ms], level=ENC_LEVEL_ONE)
    outer = pk.EncryptWithRBatch(inner, [po.rand_unit(n, rng) for _ in ms], level=ENC_LEVEL_TWO)
    layer = sk.DecryptBatch(outer, level=ENC_LEVEL_TWO)        # DecryptNestedCiphertextLayer
    assert layer == inner
    assert sk.DecryptBatch(layer, level=ENC_LEVEL_ONE) == ms   # NestedDecrypt
    assert [po.nested_decrypt(sk_o, po.Ciphertext(c, po.ENC_LEVEL_TWO)) for c in outer[:3]] == ms[:3]
    # NestedAdd (operations.go:121-127): ConstMult of a level-two ciphertext by a level-one ciphertext value
    other = pk.EncryptWithRBatch([5] * 3, [po.rand_unit(n, rng) for _ in range(3)])
    summed = pk.ConstMultBatch(outer[:3], other, level=ENC_LEVEL_TWO)
    assert summed == [po.nested_add(sk_o, po.Ciphertext(a, po.ENC_LEVEL_TWO), po.Ciphertext(b)).C
                      for a, b in zip(outer[:3], other)]


@pytest.mark.parametrize("level", [0, 1])
def test_alt_encrypt_fixed_base_comb(ctx, level):
    """AltEncryptWithRAtLevel (paillier.go:221-238): c = G^m * h_s^(r mod K); the engine evaluates h_s^r with a
    fixed-base comb table.  r is deliberately wider than K so that the reference's in-place r.Mod(r, K) matters."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(1024, 1)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1, H=sk_o.H, K=sk_o.K)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(31 + level)
    nsq = n if level == 0 else n * n
    ms = [0, 1, nsq - 1] + [rng.randrange(nsq) for _ in range(17)]
    rs = [rng.randrange(n) for _ in ms]
    rs[0], rs[1], rs[2] = 0, sk_o.K, sk_o.K - 1
    cts, rred = pk.AltEncryptWithRBatch(ms, rs, level=level)
    want = [po.alt_encrypt_with_r_at_level(sk_o, m, r, level) for m, r in zip(ms, rs)]
    assert cts == [w[0].C for w in want]
    assert rred == [w[1] for w in want]
    assert sk.DecryptBatch(cts, level=level) == ms


def test_generator_other_than_n_plus_1_and_nested_sub(ctx):
    """PublicKey.G is an exported field (paillier.go:48): with G != N+1 the engine must evaluate the literal
    Exp(G, m, n^2).  Also NestedSub (operations.go:130-140)."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(1024, 1)
    n = sk_o.N
    rng = random.Random(8)
    g = (1 + n) * pow(rng.randrange(2, n), n, n * n) % (n * n)          # another valid generator, not n + 1
    pk_o = po.PublicKey(N=n, G=g)
    pk = pa.PublicKey(ctx, n, g)
    ms = [0, 1, n - 1] + [rng.randrange(n) for _ in range(7)]
    rs = [po.rand_unit(n, rng) for _ in ms]
    assert pk.EncryptWithRBatch(ms, rs) == [po.encrypt_with_r(pk_o, m, r).C for m, r in zip(ms, rs)]
    # NestedSub on the standard generator
    pk1 = pa.PublicKey(ctx, n, n + 1)
    inner = [po.encrypt_with_r(sk_o, m, r).C for m, r in zip(ms[:4], rs[:4])]
    outer = [po.encrypt_with_r_at_level(sk_o, c, r, po.ENC_LEVEL_TWO).C for c, r in zip(inner, rs[4:8])]
    other = [po.encrypt_with_r(sk_o, 3, r).C for r in rs[:4]]
    assert pk1.NestedSubBatch(outer, other) == [
        po.nested_sub(sk_o, po.Ciphertext(a, po.ENC_LEVEL_TWO), po.Ciphertext(b)).C for a, b in zip(outer, other)]


def test_host_randomness_forms(ctx):
    """Encrypt / NestedEncrypt / Randomize draw r on the host (as the reference does with crypto/rand); results are checked
    by decryption, the property the reference's own randomized tests check (paillier_test.go:52-87, operations_test.go:92-110)."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(1024, 1)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(4)
    ms = [rng.randrange(n) for _ in range(30)] + [0, 1]
    cts = pk.EncryptBatch(ms)
    assert sk.DecryptBatch(cts) == ms and len(set(cts)) == len(cts)
    rer = pk.RandomizeBatch(cts)
    assert rer != cts and sk.DecryptBatch(rer) == ms
    nested = pk.NestedEncryptBatch(ms)
    assert sk.NestedDecryptBatch(nested) == ms
    assert sk.NestedDecryptBatch(nested[:2]) == [po.nested_decrypt(sk_o, po.Ciphertext(c, po.ENC_LEVEL_TWO)) for c in nested[:2]]


@pytest.mark.parametrize("bits", ["1024", "2048"])
def test_level_two_encrypt_lifts_through_n_squared(ctx, bits):
    """Level-two EncryptWithR computes r^(n^2) mod n^3 (paillier.go:213) as ((r mod n^2)^n mod n^2)^n mod n^3 -- an identity
    for EVERY integer r (x = x' mod n^k implies x^n = x'^n mod n^(k+1)), units or not, reduced or not.  With the lift switched
    off the literal exponentiation runs; both must give the oracle's integers."""
    import paillier_amd as pa
    from paillier_amd import ENC_LEVEL_TWO
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"][bits]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n = p * q
    n2, n3 = n * n, n ** 3
    rng = random.Random(int(bits) + 9)
    many = 280 if int(bits) <= 2048 else 40          # (the oracle's 9 216-bit powers are slow)
    rs = [1, 2, n - 1, p, q, 3 * p, n2 - 1, n + 5, n2 - n - 1] + [rng.randrange(1, n) for _ in range(many)] + [rng.randrange(n2) for _ in range(8)]
    ms = [rng.randrange(n2) for _ in rs]
    pk = pa.PublicKey(ctx, n, n + 1)
    sk_o = po.PublicKey(N=n, G=n + 1)
    got = {}
    try:
        for flag in (1, 0):
            ctx.set_flag("lift", flag)
            got[flag] = pk.EncryptWithRBatch(ms, rs, level=ENC_LEVEL_TWO)
    finally:
        ctx.set_flag("lift", 1)
    assert got[1] == got[0]
    # no launch of the call fell back to the compiler-generated kernel -- also at 3072 bits, whose n^3 (9 216 bits) runs on the
    # eight-lane shape vm_asm_42_8 (round 3; the (83,4) shape of rounds 1 and 2 existed in the hipcc kernel only)
    assert ctx.last_vm_launches() > 0 and ctx.last_vm_asm() == ctx.last_vm_launches()
    idx = list(range(12)) + list(range(len(rs) - 8, len(rs)))
    assert [got[1][i] for i in idx] == [po.encrypt_with_r_at_level(sk_o, ms[i], rs[i], po.ENC_LEVEL_TWO).C for i in idx]
    assert [got[1][i] for i in idx] == [pow(rs[i], n2, n3) * (1 + ms[i] * n + ms[i] * (ms[i] - 1) // 2 * n2) % n3 for i in idx]


def test_alt_encrypt_and_constant_forms_with_library_randomness(ctx):
    """AltEncryptAtLevel (paillier.go:244-255), EncryptZero / EncryptOne (+AtLevel, :272-289): fresh randomness from the
    library, so the check is decryption (paillier_test.go:100-156)."""
    import paillier_amd as pa
    from paillier_amd import ENC_LEVEL_ONE, ENC_LEVEL_TWO
    sk_o, p, q = po.keygen_seeded(1024, 1)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1, sk_o.H, sk_o.K)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(5)
    for level, bound in ((ENC_LEVEL_ONE, n), (ENC_LEVEL_TWO, n * n)):
        ms = [0, 1, bound - 1] + [rng.randrange(bound) for _ in range(5)]
        a, b = pk.AltEncryptBatch(ms, level), pk.AltEncryptBatch(ms, level)
        assert a != b and sk.DecryptBatch(a, level=level) == ms == sk.DecryptBatch(b, level=level)
        assert sk.DecryptBatch(pk.EncryptZeroBatch(3, level), level=level) == [0, 0, 0]
        assert sk.DecryptBatch(pk.EncryptOneBatch(3, level), level=level) == [1, 1, 1]


def test_per_ciphertext_constants_beyond_the_seven_bit_tables(ctx):
    """Level-two ConstMult with one constant per ciphertext at 40 960 ciphertexts: the 128-entry window tables of the three-digit
    kernel would not fit its 32-bit gather offsets, so the ladder takes 5-bit windows -- on number-major tables (VM_STORET /
    VM_MULVT5), like the 7-bit ones.  Sampled lanes against pow(); the same ciphertexts in two halves (7-bit windows) give the
    same integers everywhere."""
    import paillier_amd as pa
    from paillier_amd import ENC_LEVEL_TWO
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    n = int(k["n"], 16)
    n3 = n ** 3
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(4096)
    B = 40960
    base = [rng.randrange(1, n3) for _ in range(64)]
    cts = [base[i % 64] + i for i in range(B)]
    ks = [rng.getrandbits(192) for _ in range(B)]       # short constants keep the run short; windows above them are zero
    ks[0], ks[1] = 0, 1
    got = pk.ConstMultBatch(cts, ks, level=ENC_LEVEL_TWO)
    assert ctx.last_vm_asm() == ctx.last_vm_launches()
    for i in (0, 1, 2, 77, 20479, 20480, B - 1):
        assert got[i] == pow(cts[i], ks[i], n3), i
    h = B // 2
    assert pk.ConstMultBatch(cts[:h], ks[:h], level=ENC_LEVEL_TWO) + pk.ConstMultBatch(cts[h:], ks[h:], level=ENC_LEVEL_TWO) == got


@pytest.mark.parametrize("bits", ["1024", "2048", "3072"])
def test_key_holder_level_two_encrypt_by_the_lift(ctx, bits):
    """sk.EncryptWithRAtLevel(m, r, 2) (SecretKey embeds PublicKey, paillier.go:59-62,206-218): the key holder computes r^(n^2) mod n^3
    as the Teichmueller lift of (r mod p)^(q^2 mod (p - 1)) (and likewise modulo q^3) -- the same integers as the public path and
    the oracle for every unit r, for r >= n and m >= n^2; a batch with an r that shares a factor with n takes the public path and
    still gets the reference's integers; flag "lift" 0 is the public path outright."""
    import paillier_amd as pa
    from paillier_amd import ENC_LEVEL_TWO
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"][bits]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    n2, n3 = n * n, n ** 3
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    rng = random.Random(int(bits) + 77)
    ms = [0, 1, n - 1, n, n2 - 1, n2 + 5] + [rng.randrange(n2) for _ in range(270)]
    rs = [1, n - 1, 2, n + 7, po.rand_unit(n, rng), po.rand_unit(n, rng)] + [po.rand_unit(n, rng) for _ in range(270)]
    want = pk.EncryptWithRBatch(ms, rs, level=ENC_LEVEL_TWO)
    got = sk.EncryptWithRBatch(ms, rs, level=ENC_LEVEL_TWO)
    assert got == want
    assert got[:8] == [pow(r, n2, n3) * pow(n + 1, m, n3) % n3 for m, r in zip(ms[:8], rs[:8])]
    assert got[6:10] == [po.encrypt_with_r_at_level(sk_o, m, r, po.ENC_LEVEL_TWO).C for m, r in zip(ms[6:10], rs[6:10])]
    launches_lift = ctx.last_vm_launches()
    assert sk.DecryptBatch(got[:40], level=ENC_LEVEL_TWO) == [m % n2 for m in ms[:40]]
    # non-units among the r: the reference's formula verbatim
    rs2 = list(rs)
    rs2[3], rs2[9] = p, 5 * q
    got2 = sk.EncryptWithRBatch(ms, rs2, level=ENC_LEVEL_TWO)
    assert got2 == pk.EncryptWithRBatch(ms, rs2, level=ENC_LEVEL_TWO)
    assert got2[3] == pow(p, n2, n3) * pow(n + 1, ms[3], n3) % n3
    ctx.set_flag("lift", 0)
    try:
        assert sk.EncryptWithRBatch(ms[:20], rs[:20], level=ENC_LEVEL_TWO) == want[:20]
    finally:
        ctx.set_flag("lift", 1)
    assert launches_lift >= 2


@pytest.mark.parametrize("level", [0, 1])
def test_seven_bit_comb_asm_and_hipcc_kernels_agree(ctx, level):
    """VM_MULCV7 (the fixed-base comb with 7-bit windows: AltEncrypt's h_s^r, the share ZKP's V^r / V^Z): the hand-assembled kernels
    and the hipcc twin give the same ciphertexts (and the oracle's), at a batch that is re-sliced over more lanes and at one that is not."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(2048, 9)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1, H=sk_o.H, K=sk_o.K)
    rng = random.Random(131 + level)
    nsq = n if level == 0 else n * n
    ms = [rng.randrange(nsq) for _ in range(300)]
    rs = [0, 1, sk_o.K - 1, sk_o.K + 3] + [rng.getrandbits(1100) for _ in range(296)]
    got = {}
    for asm in (1, 0):
        ctx.set_flag("asm", asm)
        try:
            got[asm] = pk.AltEncryptWithRBatch(ms, rs, level=level)
            assert (ctx.last_vm_asm() > 0) == bool(asm)
        finally:
            ctx.set_flag("asm", 1)
    assert got[1] == got[0]
    want = [po.alt_encrypt_with_r_at_level(sk_o, m, r, level) for m, r in zip(ms[:6], rs[:6])]
    assert got[1][0][:6] == [w[0].C for w in want] and got[1][1][:6] == [w[1] for w in want]


def test_crt_ladders_modulo_the_cubes_on_two_lanes_per_digit(ctx):
    """Round 5: the ladders modulo p^3, q^3 of a small batch -- level-two Decrypt, the Teichmueller lifts and per-statement plaintexts of
    the DDLEQ prover -- run on the three-digit kernel with TWO lanes per digit (vm_asm_19_112: the digits of the 37-limb primes in two
    slices of 19 limbs, radix changed on the way in and out; plan::crt_triple_lanes6, flag "lanes8").  Same integers as one lane per
    digit (lanes8 0), as the oracle, non-units included; the profile shows that the flag changed the kernel."""
    import paillier_amd as pa
    from paillier_amd import ENC_LEVEL_TWO
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    n2, n3 = n * n, n ** 3
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    rng = random.Random(4242)
    ms = [0, 1, n - 1, n, n2 - 1] + [rng.randrange(n2) for _ in range(35)]
    rs = [po.rand_unit(n, rng) for _ in ms]
    cts = pk.EncryptWithRBatch(ms, rs, level=ENC_LEVEL_TWO)
    weird = [rng.randrange(n3) for _ in range(3)] + [n3 - 1, 1, p, n, n2, 0, 5 * n2, q * q]
    ms1 = [rng.randrange(n) for _ in range(6)]
    got, kern = {}, {}
    try:
        for lanes8 in (1, 0):
            ctx.set_flag("lanes8", lanes8)
            dec = sk.DecryptBatch(cts, level=ENC_LEVEL_TWO)
            kern[lanes8] = ctx.last_profile()["kernel"]
            assert ctx.last_vm_asm() == ctx.last_vm_launches()
            w, st = sk.DecryptBatch(weird, level=ENC_LEVEL_TWO, return_status=True)
            e2 = sk.EncryptWithRBatch(ms1, rs[:len(ms1)], ENC_LEVEL_TWO)          # the key holder's form: a lift modulo p^3, q^3
            got[lanes8] = (dec, w, st.tolist(), e2)
    finally:
        ctx.set_flag("lanes8", 1)
    assert got[1] == got[0]
    assert kern[1] == "vm_asm_19_112" and kern[0] == "vm_asm_37_48", kern
    assert got[1][0] == ms
    assert got[1][1] == [po.decrypt(sk_o, po.Ciphertext(c, po.ENC_LEVEL_TWO)) for c in weird]
    assert got[1][3] == [po.encrypt_with_r_at_level(sk_o, m, r, po.ENC_LEVEL_TWO).C for m, r in zip(ms1, rs)]
