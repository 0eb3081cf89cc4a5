"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the same
seeded inputs.  Bit-exact (integer work): every comparison is `==` on Python ints."""
import random

import pytest

from oracle import paillier_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


def rand_odd(bits, rng):
    return rng.getrandbits(bits) | (1 << (bits - 1)) | 1


# (modulus bits) -> kernel shape exercised: 1024 (37,1) 1536 (55,1) 2048 (74,1) 3072 (55,2) 4096 (74,2) 6144 (55,4) 9216 (42,8)
@pytest.mark.parametrize("bits", [1024, 1536, 2048, 3072, 4096, 6144, 9216, 521, 2069, 4141])
def test_modmul_matches_python(ctx, bits):
    import paillier_amd as pa
    rng = random.Random(bits)
    n = rand_odd(bits, rng)
    mod = pa.Modulus(ctx, n)
    a = [rng.randrange(n) for _ in range(300)] + [0, 1, n - 1, n - 1]
    b = [rng.randrange(n) for _ in range(300)] + [5, n - 1, n - 1, 1]
    got = mod.mul_batch(a, b)
    assert got == [x * y % n for x, y in zip(a, b)]


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096, 6144, 9216])
def test_modexp_shared_matches_python(ctx, bits):
    import paillier_amd as pa
    rng = random.Random(bits + 1)
    n = rand_odd(bits, rng)
    mod = pa.Modulus(ctx, n)
    bases = [rng.randrange(n) for _ in range(70 if bits <= 4096 else 24)] + [0, 1, n - 1]       # (the oracle's wide powers are the slow part)
    for e in (rng.getrandbits(bits // 2), 0, 1, 2, 31, 32, rng.getrandbits(200) << 77):
        got = mod.exp_batch(bases, e)
        assert got == [po.gmp_exp(x, e, n) for x in bases], f"e={e:#x}"


def test_modexp_wide_base(ctx):
    """mpz_powm accepts base >= modulus; the VM reduces a double-width base with a Horner prologue."""
    import paillier_amd as pa
    rng = random.Random(5)
    n = rand_odd(2048, rng)
    mod = pa.Modulus(ctx, n)
    bases = [rng.getrandbits(4096) for _ in range(40)] + [n, n * n - 1, (1 << 4096) - 1]
    e = rng.getrandbits(300)
    assert mod.exp_batch(bases, e, base_bytes=512) == [pow(x, e, n) for x in bases]


@pytest.mark.parametrize("bits", [2048, 4096])
def test_modexp_per_lane_exponent(ctx, bits):
    import paillier_amd as pa
    rng = random.Random(bits + 2)
    n = rand_odd(bits, rng)
    mod = pa.Modulus(ctx, n)
    bases = [rng.randrange(n) for _ in range(66)]
    exps = [rng.getrandbits(rng.choice([1, 17, 64, 300, 511])) for _ in bases]
    exps[0], exps[1] = 0, 1
    assert mod.exp_batch(bases, exps) == [po.gmp_exp(x, e, n) for x, e in zip(bases, exps)]


@pytest.fixture(scope="module")
def key1024():
    sk, p, q = po.keygen_seeded(1024, 1)
    return sk, p, q


def test_config1_encrypt_decrypt_roundtrip_1024(ctx, key1024):
    """BASELINE config 1: 1024-bit key, 256 EncryptWithR -> Decrypt round trips (paillier_test.go:52-63,158-173)."""
    import paillier_amd as pa
    sk_o, p, q = key1024
    rng = random.Random(1)
    pk = pa.PublicKey(ctx, sk_o.N, sk_o.G)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    assert sk.has_crt
    ms = [rng.randrange(sk_o.N) for _ in range(252)] + [0, 1, sk_o.N - 1, 12]
    rs = [po.rand_unit(sk_o.N, rng) for _ in ms]
    cts = pk.EncryptWithRBatch(ms, rs)
    assert cts == [po.encrypt_with_r(sk_o, m, r).C for m, r in zip(ms, rs)]
    got, status = sk.DecryptBatch(cts, return_status=True)
    assert got == ms
    assert got == [po.decrypt(sk_o, po.Ciphertext(c)) for c in cts[:8]] + ms[8:]
    assert not status.any()


def test_add_constmult_1024(ctx, key1024):
    import paillier_amd as pa
    sk_o, p, q = key1024
    rng = random.Random(3)
    pk = pa.PublicKey(ctx, sk_o.N, sk_o.G)
    n2 = sk_o.N ** 2
    a = [rng.randrange(n2) for _ in range(100)]
    b = [rng.randrange(n2) for _ in range(100)]
    assert pk.AddBatch(a, b) == [po.add(sk_o, po.Ciphertext(x), po.Ciphertext(y)).C for x, y in zip(a, b)]
    k = 50 ** 50 % sk_o.N  # operations_test.go:174-185
    assert pk.ConstMultBatch(a, k) == [po.const_mult(sk_o, po.Ciphertext(x), k).C for x in a]
    ks = [rng.randrange(sk_o.N) for _ in a]
    assert pk.ConstMultBatch(a[:40], ks[:40]) == [po.const_mult(sk_o, po.Ciphertext(x), kk).C for x, kk in zip(a, ks[:40])]


@pytest.mark.parametrize("bits", [1024, 1536, 2048, 3072, 4096, 6144, 8192, 9216, 9400])
def test_asm_kernel_matches_compiler_kernel(ctx, bits):
    """The hand-scheduled assembly VM kernel and the hipcc-generated one must agree bit for bit (and with Python)."""
    import paillier_amd as pa
    rng = random.Random(bits + 7)
    n = rand_odd(bits, rng)
    mod = pa.Modulus(ctx, n)
    bases = [rng.randrange(n) for _ in range(130)] + [0, 1, n - 1]
    e = rng.getrandbits(160) | 1
    try:
        ctx.set_flag("asm", 1)
        got_asm = mod.exp_batch(bases, e)
        assert ctx.last_vm_asm() >= 1, "assembly kernel was not used"
        ctx.set_flag("asm", 0)
        got_cc = mod.exp_batch(bases, e)
        assert ctx.last_vm_asm() == 0
    finally:
        ctx.set_flag("asm", 1)
    assert got_asm == got_cc == [pow(b, e, n) for b in bases]


@pytest.mark.parametrize("bits,count", [(1024, 1), (2048, 5), (2048, 300), (4096, 1000), (6144, 257)])
def test_modinv_batch(ctx, bits, count):
    """gmp.Int.ModInverse for a batch (tree-structured Montgomery trick; sizes that are not powers of two)."""
    import paillier_amd as pa
    rng = random.Random(bits * 3 + count)
    p1, p2 = po.gen_prime_3mod4(bits // 2, rng), po.gen_prime_3mod4(bits - bits // 2, rng)
    n = p1 * p2
    mod = pa.Modulus(ctx, n)
    xs = [po.rand_unit(n, rng) for _ in range(count)]
    xs[0] = 1
    xs[-1] = n - 1
    assert mod.inv_batch(xs) == [po.gmp_mod_inverse(x, n) for x in xs]
    if count >= 5:
        bad = list(xs)
        bad[3] = p1  # not a unit: mpz_invert is undefined; the engine reports it instead of returning garbage
        with pytest.raises(pa.PaillierHipError) as ei:
            mod.inv_batch(bad)
        assert ei.value.code == -5


def test_sub_1024(ctx, key1024):
    import paillier_amd as pa
    sk_o, p, q = key1024
    rng = random.Random(11)
    pk = pa.PublicKey(ctx, sk_o.N, sk_o.G)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    ms1 = [rng.randrange(sk_o.N) for _ in range(70)]
    ms2 = [rng.randrange(sk_o.N) for _ in range(70)]
    c1 = pk.EncryptWithRBatch(ms1, [po.rand_unit(sk_o.N, rng) for _ in ms1])
    c2 = pk.EncryptWithRBatch(ms2, [po.rand_unit(sk_o.N, rng) for _ in ms2])
    got = pk.SubBatch(c1, c2)
    assert got == [po.sub(sk_o, po.Ciphertext(a), po.Ciphertext(b)).C for a, b in zip(c1, c2)]
    assert sk.DecryptBatch(got) == [(a - b) % sk_o.N for a, b in zip(ms1, ms2)]  # operations_test.go:52-70


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096, 6144, 9216])
def test_asm_kernel_per_number_exponent(ctx, bits):
    """VM_MULV (per-number 4-bit windows, table gather) in the assembly kernel vs the hipcc kernel vs Python."""
    import paillier_amd as pa
    rng = random.Random(bits + 9)
    n = rand_odd(bits, rng)
    mod = pa.Modulus(ctx, n)
    bases = [rng.randrange(n) for _ in range(200)]
    exps = [rng.getrandbits(rng.choice([3, 28, 29, 57, 200])) for _ in bases]
    try:
        ctx.set_flag("asm", 1)
        got_asm = mod.exp_batch(bases, exps)
        assert ctx.last_vm_asm() >= 1, "assembly kernel was not used"
        ctx.set_flag("asm", 0)
        got_cc = mod.exp_batch(bases, exps)
    finally:
        ctx.set_flag("asm", 1)
    assert got_asm == got_cc == [po.gmp_exp(b, e, n) for b, e in zip(bases, exps)]


@pytest.mark.parametrize("count", [1, 255, 257, 1000])
def test_ragged_batch_sizes(ctx, key1024, count):
    """Batches that are not multiples of the 256-lane block (padding lanes must never leak into results)."""
    import paillier_amd as pa
    sk_o, p, q = key1024
    n = sk_o.N
    rng = random.Random(count)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    ms = [rng.randrange(n) for _ in range(count)]
    rs = [po.rand_unit(n, rng) for _ in range(count)]
    cts = pk.EncryptWithRBatch(ms, rs)
    assert cts[:2] == [po.encrypt_with_r(sk_o, m, r).C for m, r in zip(ms[:2], rs[:2])]
    assert cts[-1] == po.encrypt_with_r(sk_o, ms[-1], rs[-1]).C
    assert sk.DecryptBatch(cts) == ms
    doubled = pk.AddBatch(cts, cts)
    assert sk.DecryptBatch(doubled) == [(2 * m) % n for m in ms]
    assert pk.SubBatch(doubled, cts) == cts
    assert sk.DecryptBatch(pk.ConstMultBatch(cts, 3)) == [(3 * m) % n for m in ms]


def test_empty_and_bad_arguments(ctx, key1024):
    import numpy as np
    import paillier_amd as pa
    sk_o, p, q = key1024
    pk = pa.PublicKey(ctx, sk_o.N, sk_o.N + 1)
    with pytest.raises(pa.PaillierHipError):     # empty batch
        pk.encrypt_with_r_raw(0, np.zeros((1, 128), np.uint8), 128, np.zeros((1, 128), np.uint8), 128,
                              np.zeros((1, 256), np.uint8), 256)
    with pytest.raises(pa.PaillierHipError):     # even modulus
        pa.Modulus(ctx, 1 << 1024)
    with pytest.raises(pa.PaillierHipError):     # modulus wider than the built kernels
        pa.Modulus(ctx, (1 << 12000) + 1)


@pytest.mark.parametrize("bits", [64, 128, 256])
def test_fresh_small_keys_properties(ctx, bits):
    """The reference's randomised tests draw a fresh 64-bit key per iteration (paillier_test.go:52-138,
    operations_test.go:11-128): round trips and homomorphic identities for several fresh small keys, both levels."""
    import paillier_amd as pa
    for it in range(6):
        sk_o, p, q = po.keygen_seeded(bits, 1000 * bits + it)
        n = sk_o.N
        rng = random.Random(it)
        pk = pa.PublicKey(ctx, n, n + 1)
        sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
        ms = [rng.randrange(n) for _ in range(9)] + [0, n - 1]
        rs = [po.rand_unit(n, rng) for _ in ms]
        cts = pk.EncryptWithRBatch(ms, rs)
        assert cts == [po.encrypt_with_r(sk_o, m, r).C for m, r in zip(ms, rs)]
        assert sk.DecryptBatch(cts) == ms
        assert sk.DecryptBatch(pk.AddBatch(cts, cts[::-1])) == [(a + b) % n for a, b in zip(ms, ms[::-1])]
        assert sk.DecryptBatch(pk.SubBatch(cts, cts[::-1])) == [(a - b) % n for a, b in zip(ms, ms[::-1])]
        k = rng.randrange(n)
        assert sk.DecryptBatch(pk.ConstMultBatch(cts, k)) == [m * k % n for m in ms]
        c2 = pk.EncryptWithRBatch(cts, rs, level=pa.ENC_LEVEL_TWO)            # nested (paillier_test.go:65-87)
        assert c2 == [po.encrypt_with_r_at_level(sk_o, c, r, po.ENC_LEVEL_TWO).C for c, r in zip(cts, rs)]
        assert sk.NestedDecryptBatch(c2) == ms


@pytest.mark.parametrize("bits", [4096])
def test_wave_sliced_kernel(ctx, bits):
    """Moduli of 148 limbs run on the wave-sliced kernel (slices of a number in different waves, LDS rings between them)
    when the batch is large; force that shape for a small batch and check products, squarings (symmetric schedule),
    shared-exponent and per-number-exponent ladders against Python and against the compiler-generated kernel."""
    import paillier_amd as pa
    rng = random.Random(bits + 99)
    n = rand_odd(bits, rng)
    mod = pa.Modulus(ctx, n)
    xs = [rng.randrange(n) for _ in range(380)] + [0, 1, n - 1, n - 2]
    ys = [rng.randrange(n) for _ in xs]
    e = rng.getrandbits(300) | 1
    es = [rng.getrandbits(200) for _ in xs]
    try:
        ctx.set_flag("lanes_wanted", 1)
        assert mod.mul_batch(xs, ys) == [a * b % n for a, b in zip(xs, ys)]
        assert ctx.last_vm_asm() >= 1
        got = mod.exp_batch(xs, e)
        assert got == [pow(a, e, n) for a in xs]
        assert mod.exp_batch(xs, es) == [pow(a, k, n) for a, k in zip(xs, es)]
        ctx.set_flag("asm", 0)
        assert mod.exp_batch(xs, e) == got
    finally:
        ctx.set_flag("asm", 1)
        ctx.set_flag("lanes_wanted", 0)


@pytest.mark.parametrize("count", [1, 15, 17, 33, 2047, 2049, 4095, 4097, 8191, 8193])
def test_batch_sizes_around_the_latency_kernels_limits(ctx, count):
    """Round 5: 2048-bit keys pick their kernels by batch size -- sixteen lanes per number for r^n modulo n^2 up to 2 048 numbers, eight
    for the CRT halves up to 8 192, four for the ladders modulo the primes, two lanes per digit modulo p^3, q^3 up to 4 096 (DESIGN.md
    section 5) -- with blocks of 16 ... 64 numbers.  Counts that are no multiple of any block and counts on both sides of every limit:
    Encrypt (public and key holder, both levels) -> Decrypt round trips, the first and last ciphertexts against the oracle, and the
    same calls with the wide kernels switched off."""
    import json, os
    import paillier_amd as pa
    from paillier_amd import ENC_LEVEL_TWO
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keys.json")))["paillier"]["2048"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    rng = random.Random(count)
    ms = [rng.randrange(n) for _ in range(count)]
    rs = [po.rand_unit(n, rng) for _ in range(count)]
    cts = pk.EncryptWithRBatch(ms, rs)
    for i in (0, count - 1):
        assert cts[i] == po.encrypt_with_r(sk_o, ms[i], rs[i]).C
    assert sk.EncryptWithRBatch(ms, rs) == cts                      # the key holder's form: through the primes and a lift
    assert sk.DecryptBatch(cts) == ms
    c2 = min(count, 4200)                                           # (level two: three times the bytes per number)
    ct2 = sk.EncryptWithRBatch(cts[:c2], rs[:c2], ENC_LEVEL_TWO)
    assert ct2[-1] == po.encrypt_with_r_at_level(sk_o, cts[c2 - 1], rs[c2 - 1], po.ENC_LEVEL_TWO).C
    assert sk.DecryptBatch(ct2, level=ENC_LEVEL_TWO) == cts[:c2]
    if count in (17, 2049, 4097):
        try:
            for flag in ("lanes8", "lanes16", "prime_lanes"):
                ctx.set_flag(flag, 0)
            assert pk.EncryptWithRBatch(ms, rs) == cts and sk.EncryptWithRBatch(ms, rs) == cts and sk.DecryptBatch(cts) == ms
            assert sk.DecryptBatch(ct2, level=ENC_LEVEL_TWO) == cts[:c2]
        finally:
            for flag in ("lanes8", "lanes16", "prime_lanes"):
                ctx.set_flag(flag, 1)
