"""The three-digit kernel (gen_vm_asm.GenQ3, vm_asm_<H>_48): residues modulo n^3 as a0 + a1 n + a2 n^2 in the lanes of a quad.

1. Every digit the kernel produces is an exact integer function of its inputs (block-wise Montgomery reductions whose
   quotient digits feed the next digit): compared bit for bit with a few lines of Python, and the value identity
   value(x) * value(y) = value(result) * R (mod n^3) is checked on top.
2. The paths that use it -- level-two EncryptWithR, ConstMult / NestedAdd, the DDLEQ verifier -- give identical integers with
   the kernel switched off (the 3H-limb kernels) and equal to the oracle / the committed fixtures."""
import json
import os
import random

import numpy as np
import pytest

from oracle import paillier_oracle as po

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LB = 28
OPS = dict(END=0, LOAD=1, STORE=2, SQR=4, MUL=5)


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


def limbs(v, w):
    return [(v >> (LB * i)) & ((1 << LB) - 1) for i in range(w)]


def value(arr):
    return sum(int(x) << (LB * i) for i, x in enumerate(arr))


@pytest.mark.parametrize("bits,H,lanes", [(1024, 37, 3), (2048, 74, 3), (2048, 74, 6), (3072, 110, 6)])
def test_digit_model(ctx, bits, H, lanes):
    # lanes = 3: GenQ3, one lane per digit (+ a helper lane); lanes = 6: GenQ6, two lanes per digit (vm_asm_37_112, vm_asm_55_112:
    # digits of 74 / 110 limbs), whose product is the two-pass form at every width
    rng = random.Random(bits)
    n = po.gen_prime_3mod4(bits // 2, rng) * po.gen_prime_3mod4(bits // 2, rng)
    R = 1 << (LB * H)
    nb = 256
    D = sum(1 << (LB * (j + 1)) for j in range(H))
    k1 = D // n + 1
    C1 = k1 * n
    C2 = D + ((-k1 - D) % n)
    # the kernel streams these as the limbs e_j + 2^28 of C = D + E, D = sum_j 2^(28 j + 28), 0 <= E < 2^(28 H): every word
    # exceeds any quotient digit m_j < 2^28
    assert 0 < C1 - D <= n and 0 <= C2 - D < n and C1 % n == 0 and (C2 + k1) % n == 0
    nprime = (-pow(n, -1, R)) % R

    def redc(P):
        m = (P * nprime) % R
        return m, (P + m * n) // R

    xs = [[rng.randrange(2 * n) for _ in range(3)] for _ in range(nb)]
    ys = [[rng.randrange(2 * n) for _ in range(3)] for _ in range(nb)]
    xs[0], ys[0] = [0, 0, 0], [1, 2, 3]
    xs[1], ys[1] = [n - 1, n - 1, n - 1], [2 * n - 1] * 3
    mem = np.zeros((4, 3 * H, nb), dtype=np.uint32)
    for g in range(nb):
        for d in range(3):
            mem[0, d * H:(d + 1) * H, g] = limbs(xs[g][d], H)
            mem[1, d * H:(d + 1) * H, g] = limbs(ys[g][d], H)
    prog = [OPS["LOAD"], 0, OPS["SQR"], 0, OPS["STORE"], 2, OPS["LOAD"], 0, OPS["MUL"], 1, OPS["STORE"], 3, OPS["END"], 0]
    out, _, h = ctx.pair_debug_run(n, prog, mem, 4, nb, lanes=lanes)
    assert h == H
    n3 = n ** 3
    val = lambda d: d[0] + d[1] * n + d[2] * n * n
    for g in range(nb):
        a0, a1, a2 = xs[g]
        b0, b1, b2 = ys[g]
        got_sq = [value(out[2, d * H:(d + 1) * H, g]) for d in range(3)]
        got_mul = [value(out[3, d * H:(d + 1) * H, g]) for d in range(3)]
        m00, t00 = redc(a0 * a0)
        m1, t1 = redc(2 * a0 * a1 + C1 - m00)
        _, t2 = redc(2 * a0 * a2 + C2 - m1)
        _, t3 = redc(a1 * a1)
        assert got_sq == [t00, t1, t2 + t3], f"squaring digits, number {g}"
        assert (val(got_sq) * R - val(xs[g]) ** 2) % n3 == 0
        m00, t00 = redc(a0 * b0)
        _, t02 = redc(a0 * b2)
        if H <= 55 and lanes == 3:
            # one pass (the kernel has the registers for a copy of the digit below): ONE reduction per digit of the whole
            # coefficient, its quotient digits subtracted from the next digit
            m1, t1 = redc(a1 * b0 + a0 * b1 + C1 - m00)
            _, t2 = redc(a2 * b0 + a1 * b1 + C2 - m1)
            want = [t00, t1, t2 + t02]
        else:
            m10, t10 = redc(a1 * b0 + C1 - m00)
            _, t20 = redc(a2 * b0 + C2 - m10)
            m01, t01 = redc(a0 * b1)
            _, t11 = redc(a1 * b1 + C1 - m01)
            want = [t00, t10 + t01, t20 + t02 + t11]
        assert got_mul == want, f"product digits, number {g}"
        assert (val(got_mul) * R - val(xs[g]) * val(ys[g])) % n3 == 0
        assert all(int(v) < (1 << LB) for d in range(3) for v in out[2, d * H:(d + 1) * H - 1, g])   # canonical limbs below the top


@pytest.mark.parametrize("bits", ["1024", "2048"])
def test_level_two_paths_on_and_off(ctx, bits):
    import paillier_amd as pa
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"][bits]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    n2, n3 = n * n, n ** 3
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(int(bits) + 33)
    B = 70
    ms = [0, 1, n2 - 1] + [rng.randrange(n2) for _ in range(B - 3)]
    rs = [po.rand_unit(n, rng) for _ in ms]
    ks = [rng.randrange(n2) for _ in ms]
    res = {}
    for flag in (1, 0):
        ctx.set_flag("triple", flag)
        try:
            cts = pk.EncryptWithRBatch(ms, rs, level=pa.ENC_LEVEL_TWO)
            kern = ctx.last_profile()["kernel"]
            # (a batch this small takes the two-lanes-per-digit variant of the three-digit kernel, vm_asm_<H/2>_112, or -- a lone
            # shared-exponent ladder on a 2048-bit key -- four lanes per digit, vm_asm_19_160)
            assert (kern.endswith("_48") or kern.endswith("_112") or kern.endswith("_160")) == bool(flag), kern
            res[flag] = (cts, pk.ConstMultBatch(cts, ks, level=pa.ENC_LEVEL_TWO), pk.ConstMultBatch(cts, ks[5], level=pa.ENC_LEVEL_TWO))
        finally:
            ctx.set_flag("triple", 1)
    assert res[1] == res[0]
    assert res[1][0][:6] == [po.encrypt_with_r_at_level(sk_o, m, r, po.ENC_LEVEL_TWO).C for m, r in zip(ms[:6], rs[:6])]
    assert res[1][1][:6] == [pow(c, e, n3) for c, e in zip(res[1][0][:6], ks[:6])]
    assert res[1][2][:6] == [pow(c, ks[5], n3) for c in res[1][0][:6]]


def test_ddleq_verify_2048_on_and_off(ctx):
    import paillier_amd as pa
    H_ = lambda xs: [int(x, 16) for x in xs]
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    n = int(k["n"], 16)
    pk = pa.PublicKey(ctx, n, n + 1)
    d = json.load(open(os.path.join(G, "proofs.json")))["ddleq"]
    st = [{kk: int(v, 16) for kk, v in s.items()} for s in d["statements"]]
    ins = d["instances"][:16]
    col = lambda key: [st[i["s"]][key] for i in ins]
    xs, ys = H_(i["x"] for i in ins), H_(i["y"] for i in ins)
    al, es, fs = H_(i["alpha"] for i in ins), H_(i["e"] for i in ins), H_(i["f"] for i in ins)
    wrong = [st[(i["s"] + 1) % 4]["ct2"] for i in ins]
    fs_bad = list(fs)
    fs_bad[3] ^= 1 << 4000
    # (triple, lanes8, lanes16): a batch this small takes four lanes per digit (vm_asm_19_160, a DPP row per number: round 5) unless
    # lanes16 is off -- then two lanes per digit (vm_asm_37_112, limb-major 5-bit tables) -- or lanes8 is off -- then the one-lane digits
    # with number-major 7-bit tables (vm_asm_74_48); triple off: the generic kernel
    for flag, l8, l16, suffix in ((1, 1, 1, "_160"), (1, 1, 0, "_112"), (1, 0, 1, "_48"), (0, 1, 1, None)):
        ctx.set_flag("triple", flag)
        ctx.set_flag("lanes8", l8)
        ctx.set_flag("lanes16", l16)
        try:
            assert pk.VerifyDDLEQInstancesBatch(col("ct1"), col("ct2"), xs, ys, al, es, fs) == [True] * 16
            kern = ctx.last_profile()["kernel"]
            assert kern.endswith(suffix) if suffix else not (kern.endswith("_48") or kern.endswith("_112") or kern.endswith("_160")), kern
            assert pk.VerifyDDLEQInstancesBatch(col("ct1"), wrong, xs, ys, al, es, fs) == [bool(i["verify_wrong_ct2"]) for i in ins]
            got = pk.VerifyDDLEQInstancesBatch(col("ct1"), col("ct2"), xs, ys, al, es, fs_bad)
            assert got == [i != 3 for i in range(16)]
        finally:
            ctx.set_flag("triple", 1)
            ctx.set_flag("lanes8", 1)
            ctx.set_flag("lanes16", 1)


def test_split_through_n_squared_on_and_off(ctx):
    """NestedRandomize and the DDLEQ verifier compute x^e y^(n^2) mod n^3 as (x^(e1) y^n mod n^2)^n x^(e0), e = e0 + e1 n
    (ddleq.cpp dual_pow_n3).  With pgpu_ctx_set_flag("lift", 0) the literal interleaved ladder of 4 096 squarings runs: same
    integers, same verdicts, and both equal the committed 2048-bit fixtures / pow()."""
    import paillier_amd as pa
    H_ = lambda xs: [int(x, 16) for x in xs]
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    n = int(k["n"], 16)
    n2, n3 = n * n, n ** 3
    pk = pa.PublicKey(ctx, n, n + 1)
    d = json.load(open(os.path.join(G, "proofs.json")))["ddleq"]
    st = [{kk: int(v, 16) for kk, v in s.items()} for s in d["statements"]]
    ins = d["instances"][:12]
    col = lambda key: [st[i["s"]][key] for i in ins]
    xs, ys = H_(i["x"] for i in ins), H_(i["y"] for i in ins)
    al, es, fs = H_(i["alpha"] for i in ins), H_(i["e"] for i in ins), H_(i["f"] for i in ins)
    wrong = [st[(i["s"] + 1) % 4]["ct2"] for i in ins]
    rng = random.Random(4096)
    cts = [rng.randrange(n3) for _ in range(40)] + [0, 1, n, n2, n3 - 1, 3 * n2]     # non-units and small values ride along
    a_s = [rng.randrange(1, n) for _ in cts]
    b_s = [rng.randrange(1, n) for _ in cts]
    got = {}
    for flag in (1, 0):
        ctx.set_flag("lift", flag)
        try:
            got[flag] = (pk.VerifyDDLEQInstancesBatch(col("ct1"), col("ct2"), xs, ys, al, es, fs),
                         pk.VerifyDDLEQInstancesBatch(col("ct1"), wrong, xs, ys, al, es, fs),
                         pk.NestedRandomizeWithABBatch(cts, a_s, b_s))
        finally:
            ctx.set_flag("lift", 1)
    assert got[1] == got[0]
    assert got[1][0] == [True] * len(ins)
    assert got[1][1] == [bool(i["verify_wrong_ct2"]) for i in ins]
    pick = list(range(6)) + list(range(len(cts) - 6, len(cts)))
    assert [got[1][2][i] for i in pick] == [pow(cts[i], pow(a_s[i], n, n2), n3) * pow(b_s[i], n2, n3) % n3 for i in pick]


@pytest.mark.parametrize("batch", [70, 600])
def test_pair_form_handed_to_the_digit_kernel(ctx, batch):
    """The power modulo n^2 of a lift goes to the ladder modulo n^3 as (a0, a1, 0) -- its pair digits with a zero third digit --
    instead of leaving pair form and entering digit form (flag handover = 0): level-two Encrypt (incl. r = 1, r = n - 1, whose
    power has digit a1 = 0 or small), NestedRandomize-style dual ladders through VerifyDDLEQ, both ways, same integers; the
    encryptions also against the oracle."""
    import paillier_amd as pa
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    n2 = n * n
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(batch)
    ms = [0, 1, n2 - 1] + [rng.randrange(n2) for _ in range(batch - 3)]
    rs = [1, n - 1, 2] + [po.rand_unit(n, rng) for _ in range(batch - 3)]
    H_ = lambda xs: [int(x, 16) for x in xs]
    d = json.load(open(os.path.join(G, "proofs.json")))["ddleq"]
    st = [{kk: int(v, 16) for kk, v in s.items()} for s in d["statements"]]
    ins = d["instances"][:16]
    col = lambda key: [st[i["s"]][key] for i in ins]
    xs, ys = H_(i["x"] for i in ins), H_(i["y"] for i in ins)
    al, es, fs = H_(i["alpha"] for i in ins), H_(i["e"] for i in ins), H_(i["f"] for i in ins)
    res = {}
    for flag in (1, 0):
        ctx.set_flag("handover", flag)
        try:
            res[flag] = pk.EncryptWithRBatch(ms, rs, level=pa.ENC_LEVEL_TWO)
            assert pk.VerifyDDLEQInstancesBatch(col("ct1"), col("ct2"), xs, ys, al, es, fs) == [True] * 16
        finally:
            ctx.set_flag("handover", 1)
    assert res[1] == res[0]
    assert res[1][:8] == [po.encrypt_with_r_at_level(sk_o, m, r, po.ENC_LEVEL_TWO).C for m, r in zip(ms[:8], rs[:8])]
