"""GPU parity for the pair kernel (gen_vm_asm.py GenP): residues modulo p^2 kept as two base-p digits
y R mod p^2 = a0 + a1 p, product = two Montgomery steps modulo p.  Every digit the kernel produces is an exact integer
function of its inputs, so the comparison with the Python restatement below is bit for bit; a whole ladder is then checked
against pow(); and Decrypt (which runs its ladders on this kernel for 2048-bit keys) must not depend on the "pair" switch."""
import json
import os
import random

import numpy as np
import pytest

from model28 import to_limbs, from_limbs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
END, LOAD, STORE, SQR, MUL, MULC, MULV, ADD = 0, 1, 2, 4, 5, 6, 7, 8
C_ONE = 3          # constant 1 of every modulus context: the pair (1, 0)
LB = 28
# one-lane pair kernels: GenP (37-limb primes, 2048-bit keys) and GenP2 (55-limb primes, 3072-bit keys: quotient digits in
# LDS, multiplier digits of a product streamed from memory)
ONE_LANE = [("2048", 37), ("3072", 55)]


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


def pair_model(p, H, cadj):
    R = 1 << (LB * H)
    nneg = (-pow(p, -1, R)) % R

    def mont(u):
        m = (u * nneg) % R
        return (u + m * p) // R, m

    def pmul(a, b):
        t, m = mont(a[0] * b[0])
        c1, _ = mont(a[0] * b[1] + a[1] * b[0] + cadj - m)
        return (t, c1)
    return pmul, R


@pytest.mark.parametrize("bits,H", ONE_LANE)
def test_pair_products_match_the_integer_model(ctx, bits, H):
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"][bits]
    p = int(k["p"], 16)
    rng = random.Random(20)
    nb, nslots = 512, 6
    mem = np.zeros((nslots, 2 * H, nb), dtype=np.uint32)
    vals = {}
    for s in (0, 1):
        for g in range(nb):
            hi = 2 * p if g % 3 else p            # lazy digits up to 2p as well as canonical ones
            a = (rng.randrange(hi), rng.randrange(hi))
            if g == 0:
                a = (0, 0)
            if g == 1:
                a = (p - 1, p - 1)
            vals[s, g] = a
            mem[s, :H, g] = to_limbs(a[0], H)
            mem[s, H:, g] = to_limbs(a[1], H)
    prog = [LOAD, 0, SQR, 0, STORE, 2, LOAD, 0, MUL, 1, STORE, 3, SQR, 0, SQR, 0, MUL, 0, STORE, 4,
            LOAD, 0, ADD, 1, MULC, C_ONE, STORE, 5, END, 0]
    out, consts, h = ctx.pair_debug_run(p, prog, mem, nslots, nb)
    assert h == H and from_limbs(consts[:H]) == p
    cadj = from_limbs(consts[H:])
    assert cadj % p == 0 and all((1 << LB) <= int(c) < (2 << LB) for c in consts[H:])
    pmul, R = pair_model(p, H, cadj)
    for g in range(nb):
        x, y = vals[0, g], vals[1, g]
        sq = pmul(x, x)
        xy = pmul(x, y)
        t = pmul(xy, xy)
        t = pmul(t, t)
        t = pmul(t, x)
        s1 = pmul((x[0] + y[0], x[1] + y[1]), (1, 0))        # ADD is digit-wise and lazy; the product after it normalises
        for slot, want in ((2, sq), (3, xy), (4, t), (5, s1)):
            got = (from_limbs(out[slot, :H, g]), from_limbs(out[slot, H:, g]))
            assert got == want, (g, slot)
            assert all(int(v) < (1 << LB) for v in out[slot, :, g])


@pytest.mark.parametrize("bits,H", ONE_LANE)
def test_pair_ladder_is_a_modexp(ctx, bits, H):
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"][bits]
    q = int(k["q"], 16)
    nb, nslots = 256, 3
    rng = random.Random(21)
    R = 1 << (LB * H)
    q2 = q * q
    ys = [rng.randrange(q2) for _ in range(nb)]
    mem = np.zeros((nslots, 2 * H, nb), dtype=np.uint32)
    for g, y in enumerate(ys):
        yt = y * R % q2
        mem[0, :H, g] = to_limbs(yt % q, H)
        mem[0, H:, g] = to_limbs(yt // q, H)
    e = rng.getrandbits(90) | (1 << 89)
    prog = [LOAD, 0]
    for bit in bin(e)[3:]:
        prog += [SQR, 0]
        if bit == "1":
            prog += [MUL, 0]
    prog += [STORE, 1, END, 0]
    out, _, _ = ctx.pair_debug_run(q, prog, mem, nslots, nb)
    rinv = pow(R, -1, q2)
    for g, y in enumerate(ys):
        got = (from_limbs(out[1, :H, g]) + from_limbs(out[1, H:, g]) * q) * rinv % q2
        assert got == pow(y, e, q2), g


@pytest.mark.parametrize("one_lane", [0, 1])
def test_decrypt_does_not_depend_on_the_pair_switch(ctx, one_lane):
    """one_lane = 1 forces the one-lane kernel (GenP, what a full batch runs); 0 leaves the small-batch choice (two lanes)."""
    import paillier_amd as pa
    from oracle import paillier_oracle as po
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    rng = random.Random(22)
    cts = [rng.randrange(n * n) for _ in range(700)] + [0, 1, p, q, n, n * n - 1, p * p, 5 * q]
    try:
        ctx.set_flag("lanes_wanted", 1 if one_lane else 0)
        ctx.set_flag("pair", 1)
        with_pair, st = sk.DecryptBatch(cts, return_status=True)
        ctx.set_flag("pair", 0)
        without, st0 = sk.DecryptBatch(cts, return_status=True)
    finally:
        ctx.set_flag("pair", 1)
        ctx.set_flag("lanes_wanted", 0)
    assert with_pair == without and list(st) == list(st0)
    from math import gcd
    assert [bool(v & pa.LANE_NONUNIT) for v in st] == [gcd(c, n) != 1 for c in cts]
    assert with_pair[:40] + with_pair[-8:] == [po.decrypt(sk_o, po.Ciphertext(c)) for c in cts[:40] + cts[-8:]]


@pytest.mark.parametrize("lanes", [2, 4])
def test_two_lane_pair_kernel_matches_the_integer_model(ctx, lanes):
    """GenQ (lanes = 2): N = n^2 with a public 74-limb n, digit a0 in lane 0 and a1 in lane 1; GenQ4 (lanes = 4): every
    digit sliced over two lanes.  The two-lane product adds the two partial results r1 + r2 limb-wise (a lazy digit below
    4n), so digits are compared as integers; the four-lane product reduces a1 b0 + a0 b1 + Cadj - m in one pass."""
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    n = int(k["p"], 16) * int(k["q"], 16)
    rng = random.Random(23)
    nb, nslots, H = 256, 6, 74
    mem = np.zeros((nslots, 2 * H, nb), dtype=np.uint32)
    vals = {}
    for s in (0, 1):
        for g in range(nb):
            hi = 2 * n if g % 3 else n
            a = (rng.randrange(hi), rng.randrange(hi))
            if g == 0:
                a = (0, 0)
            if g == 1:
                a = (n - 1, n - 1)
            vals[s, g] = a
            mem[s, :H, g] = to_limbs(a[0], H)
            mem[s, H:, g] = to_limbs(a[1], H)
    prog = [LOAD, 0, SQR, 0, STORE, 2, LOAD, 0, MUL, 1, STORE, 3, SQR, 0, SQR, 0, MUL, 0, STORE, 4, END, 0]
    out, consts, h = ctx.pair_debug_run(n, prog, mem, nslots, nb, lanes=lanes)
    assert h == H and from_limbs(consts[:H]) == n
    cadj = from_limbs(consts[H:])
    assert cadj % n == 0
    R = 1 << (LB * H)
    nneg = (-pow(n, -1, R)) % R

    def mont(u):
        m = (u * nneg) % R
        return (u + m * n) // R, m

    def psq(a):
        t, m = mont(a[0] * a[0])
        return (t, mont(2 * a[0] * a[1] + cadj - m)[0])

    def pmul(a, b):      # x = a (registers), b = the slot operand
        t, m = mont(a[0] * b[0])
        if lanes == 4:   # one pass, two multiplier streams in the lanes of digit one: a single reduction of the whole cross term
            return (t, mont(a[1] * b[0] + a[0] * b[1] + cadj - m)[0])
        r2, _ = mont(a[0] * b[1])              # two passes: r2 = a0 b1, then t = a0 b0 and r1 = a1 b0 + Cadj - m
        r1, _ = mont(a[1] * b[0] + cadj - m)
        return (t, r1 + r2)

    for g in range(nb):
        x, y = vals[0, g], vals[1, g]
        sq = psq(x)
        xy = pmul(x, y)
        t = psq(psq(xy))
        t = pmul(t, x)
        for slot, want in ((2, sq), (3, xy), (4, t)):
            got = (from_limbs(out[slot, :H, g]), from_limbs(out[slot, H:, g]))
            assert got == want, (g, slot)
    # and the value represented is the product modulo n^2
    n2 = n * n
    rinv = pow(R, -1, n2)
    for g in (2, 3, 200):
        x, y = vals[0, g], vals[1, g]
        xv, yv = (x[0] + x[1] * n) % n2, (y[0] + y[1] * n) % n2
        got = (from_limbs(out[3, :H, g]) + from_limbs(out[3, H:, g]) * n) % n2
        assert got == xv * yv * rinv % n2


@pytest.mark.parametrize("one_lane", [0, 1])
def test_decrypt_3072_on_the_pair_kernels(ctx, one_lane):
    """3072-bit keys: the CRT ladders modulo p^2, q^2 (55-limb primes) run on the one-lane pair kernel GenP2 when the batch
    fills the chip (one_lane = 1 forces it for this small batch) and on the two-lane kernel below that (one_lane = 0);
    compare both with the ordinary kernels and the oracle."""
    import paillier_amd as pa
    from oracle import paillier_oracle as po
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["3072"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    rng = random.Random(24)
    cts = [rng.randrange(n * n) for _ in range(300)] + [0, 1, p, q, n, n * n - 1, p * p, 7 * q]
    try:
        ctx.set_flag("lanes_wanted", 1 if one_lane else 0)
        ctx.set_flag("pair", 1)
        with_pair, st = sk.DecryptBatch(cts, return_status=True)
        ctx.set_flag("pair", 0)
        without, st0 = sk.DecryptBatch(cts, return_status=True)
    finally:
        ctx.set_flag("pair", 1)
        ctx.set_flag("lanes_wanted", 0)
    assert with_pair == without and list(st) == list(st0)
    from math import gcd
    assert [bool(v & pa.LANE_NONUNIT) for v in st] == [gcd(c, n) != 1 for c in cts]
    assert with_pair[:20] + with_pair[-6:] == [po.decrypt(sk_o, po.Ciphertext(c)) for c in cts[:20] + cts[-6:]]


@pytest.mark.parametrize("force_two_lanes", [0, 1])
def test_const_mult_per_ciphertext_on_the_pair_kernels(ctx, force_two_lanes):
    """ConstMult with one k per ciphertext (operations.go:58-64 element-wise; NestedAdd's shape) is a per-number-exponent
    ladder modulo n^2: table gathers (MULV) on the pair kernels -- the four-lane one for a small batch, the two-lane one
    when forced -- must agree with the ordinary kernels and with Python."""
    import paillier_amd as pa
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    n = int(k["p"], 16) * int(k["q"], 16)
    n2 = n * n
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(25 + force_two_lanes)
    cs = [rng.randrange(n2) for _ in range(260)] + [0, 1, n2 - 1]
    ks = [rng.randrange(n) for _ in cs[:-2]] + [0, 1]
    try:
        ctx.set_flag("lanes_wanted", 1 if force_two_lanes else 0)
        ctx.set_flag("pair", 1)
        got = pk.ConstMultBatch(cs, ks)
        ctx.set_flag("pair", 0)
        ref = pk.ConstMultBatch(cs, ks)
    finally:
        ctx.set_flag("pair", 1)
        ctx.set_flag("lanes_wanted", 0)
    assert got == ref == [pow(c, kk, n2) for c, kk in zip(cs, ks)]


@pytest.mark.parametrize("force_pair", [0, 1])
def test_decrypt_4096_bit_key(ctx, force_pair):
    """4096-bit keys: CRT halves modulo p^2, q^2 with 74-limb primes (148-limb moduli).  Golden vectors from the oracle
    (tests/golden/key4096.json, written by make_golden.py 4096); with force_pair the ladders run on the two-lane pair kernel
    even for this small batch."""
    import paillier_amd as pa
    k = json.load(open(os.path.join(G, "key4096.json")))
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, int(k["lambda"], 16)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    assert sk.has_crt
    ms, rs, cs = ([int(x, 16) for x in k[key]] for key in ("m", "r", "c"))
    wc, wm = ([int(x, 16) for x in k[key]] for key in ("weird_c", "weird_m"))
    try:
        ctx.set_flag("lanes_wanted", 1 if force_pair else 0)
        assert pk.EncryptWithRBatch(ms, rs) == cs
        assert sk.DecryptBatch(cs) == ms
        got, st = sk.DecryptBatch(wc, return_status=True)
        assert got == wm
        from math import gcd
        assert [bool(v & pa.LANE_NONUNIT) for v in st] == [gcd(c, n) != 1 for c in wc]
        rng = random.Random(26)
        many = [rng.randrange(n * n) for _ in range(200)]
        a = sk.DecryptBatch(many)
        assert a == sk.DecryptBatch(many, flags=pa.DECRYPT_NO_CRT)
    finally:
        ctx.set_flag("lanes_wanted", 0)


@pytest.mark.parametrize("batch,lanes_wanted", [(96, 0), (96, 2 * 96), (400, 0)])
def test_per_number_windows_number_major_and_limb_major(ctx, batch, lanes_wanted):
    """x_i^(e_i) mod n^2 with one exponent per number on the pair kernels (two and four lanes per number): the window tables
    number-major (VM_STORET / VM_MULVT, the default) and limb-major (flag nm4 = 0, VM_MULV) must give the same integers as pow();
    the DDLEQ verifier's interleaved ladder (5-bit per-number windows: VM_MULVT5) is covered through ConstMult + the proofs'
    fixtures, and here through a level-one ConstMult with per-ciphertext constants."""
    import paillier_amd as pa
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    n = int(k["n"], 16)
    n2 = n * n
    m2 = pa.Modulus(ctx, n2)
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(77 + batch)
    xs = [rng.randrange(1, n2) for _ in range(batch)]
    es = [rng.randrange(0, n) for _ in range(batch)]
    es[0], es[1], es[2] = 0, 1, n - 1
    want = [pow(x, e, n2) for x, e in zip(xs, es)]
    got = {}
    try:
        ctx.set_flag("lanes_wanted", lanes_wanted)
        for nm4 in (1, 0):
            ctx.set_flag("nm4", nm4)
            got[nm4] = m2.exp_batch(xs, es)
            assert ctx.last_profile()["kernel"].startswith("vm_asm_"), ctx.last_profile()
            got[("cm", nm4)] = pk.ConstMultBatch(xs, es)
    finally:
        ctx.set_flag("nm4", 1)
        ctx.set_flag("lanes_wanted", 0)
    assert got[1] == want and got[0] == want
    assert got[("cm", 1)] == want and got[("cm", 0)] == want
