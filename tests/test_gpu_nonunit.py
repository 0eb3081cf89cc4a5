"""Hostile / malformed inputs to the batch ModInverse users: ONE element that is not a unit modulo the modulus
must not cost the rest of the batch its answers.  The reference inverts per call (mpz_invert returns 0 and leaves the
result undefined for a non-unit; VerifyProof then simply fails and CombinePartialDecryptionsZKP drops that share,
thresholdkey.go:164-172), so the batch engine reports such lanes individually: the tree's single inversion fails, a
per-lane binary GCD finds the culprits, the tree runs again without them."""
import json
import os
import random

import numpy as np
import pytest

from oracle import paillier_oracle as po

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


@pytest.mark.parametrize("bits,count", [(1024, 7), (2048, 300), (4096, 600)])
def test_modinv_reports_non_units_per_lane(ctx, bits, count):
    import paillier_amd as pa
    rng = random.Random(bits + count)
    p1, p2 = po.gen_prime_3mod4(bits // 2, rng), po.gen_prime_3mod4(bits - bits // 2, rng)
    n = p1 * p2
    mod = pa.Modulus(ctx, n)
    xs = [po.rand_unit(n, rng) for _ in range(count)]
    bad = {0: 0, 3: p1, count - 1: (p2 * rng.randrange(1, p1)) % n, count // 2: (p1 * 12345) % n}
    for i, v in bad.items():
        xs[i] = v
    got, st = mod.inv_batch(xs, return_status=True)
    for i, x in enumerate(xs):
        if i in bad:
            assert st[i] == pa.api.LANE_NOT_INVERTIBLE and got[i] == 0
        else:
            assert st[i] == 0 and got[i] == po.gmp_mod_inverse(x, n)
    with pytest.raises(pa.PaillierHipError) as ei:       # no status array: the failure is the return code
        mod.inv_batch(xs)
    assert ei.value.code == -5
    units = [x for i, x in enumerate(xs) if i not in bad]
    got, st = mod.inv_batch(units, return_status=True)    # fast path: no GCD pass, all-zero status
    assert not st.any() and got == [po.gmp_mod_inverse(x, n) for x in units]


def test_sub_with_a_non_unit_subtrahend(ctx):
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(1024, 1)
    n, n2 = sk_o.N, sk_o.N ** 2
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(5)
    a = [rng.randrange(n2) for _ in range(20)]
    b = [po.encrypt_with_r(sk_o, rng.randrange(n), po.rand_unit(n, rng)).C for _ in range(20)]
    b[4], b[11] = 0, p * 77
    got, st = pk.SubBatch(a, b, return_status=True)
    for i in range(20):
        if i in (4, 11):
            assert st[i] == pa.api.LANE_NOT_INVERTIBLE and got[i] == 0
        else:
            assert st[i] == 0 and got[i] == po.sub(sk_o, po.Ciphertext(a[i]), po.Ciphertext(b[i])).C
    with pytest.raises(pa.PaillierHipError) as ei:
        pk.SubBatch(a, b)
    assert ei.value.code == -5


@pytest.mark.parametrize("bits", ["512", "2048"])
def test_share_proofs_with_non_unit_decryptions(ctx, bits):
    """VerifyProofBatch / combine_partial_decryptions_zkp with an adversarial proof whose Decryption is 0 or a multiple of a
    prime factor of n: rejected per lane; every other proof keeps its verdict; the ZKP combine drops only that share."""
    import paillier_amd as pa
    from paillier_amd import protocols as pr
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"][bits]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    p = int(k["p"], 16)
    shares = [int(s, 16) for s in k["shares"]]
    v, vks = int(k["v"], 16), [int(x, 16) for x in k["vks"]]
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    rng = random.Random(int(bits) + 77)
    tsk0 = po.ThresholdSecretKey(N=n, G=n + 1, TotalNumberOfDecryptionServers=total, Threshold=thr, VerificationKey=v,
                                 VerificationKeys=vks, ID=1, Share=shares[0])
    ms = [rng.randrange(n) for _ in range(9)]
    cts = [po.encrypt_with_r(tsk0, m, po.rand_unit(n, rng)).C for m in ms]

    def proofs_of(sid):
        rs = [rng.randrange(n * n) for _ in cts]
        d, es, zs = tk.PartialDecryptionWithZKPBatch(sid, shares[sid - 1], v, cts, rs)
        return [pr.PartialDecryptionZKP(sid, di, ei, zi, ci) for di, ei, zi, ci in zip(d, es, zs, cts)]

    srv = [proofs_of(s) for s in (1, 2, 4, 5)]
    srv[1][2].Decryption = 0                     # server 2, ciphertext 2
    srv[1][6].Decryption = (p * 991) % (n * n)   # server 2, ciphertext 6
    s2 = srv[1]
    ok = tk.VerifyProofBatch(v, vks[1], [x.C for x in s2], [x.Decryption for x in s2], [x.E for x in s2], [x.Z for x in s2])
    assert ok == [i not in (2, 6) for i in range(len(cts))]
    assert pr.combine_partial_decryptions_zkp(tk, v, vks, srv) == ms      # three honest servers remain everywhere
    # Combine WITHOUT proofs on the same hostile share: only the two affected ciphertexts are flagged
    use = [(x[0].ID, [y.Decryption for y in x]) for x in (srv[0], srv[1], srv[3])]      # ids (1, 2, 5): lambda_2 < 0
    got, st = tk.CombinePartialDecryptionsBatch(use, return_status=True)
    assert [int(s) for s in st] == [2 if i in (2, 6) else 0 for i in range(len(cts))]
    assert [g for i, g in enumerate(got) if i not in (2, 6)] == [m for i, m in enumerate(ms) if i not in (2, 6)]
    with pytest.raises(pa.PaillierHipError) as ei:
        tk.CombinePartialDecryptionsBatch(use)
    assert ei.value.code == -5


def test_wide_strides_are_reduced_not_truncated(ctx):
    """Operands handed over with a stride wider than the modulus (leading bytes non-zero) are reduced modulo n^(s+1), as the
    reference's Exp / Mul+Mod do -- they used to be cut to the low bytes silently."""
    import ctypes as C
    import paillier_amd as pa
    from paillier_amd.api import MEM_HOST, _ptr, be_to_ints, ints_to_be
    sk_o, p, q = po.keygen_seeded(1024, 1)
    n, n2 = sk_o.N, sk_o.N ** 2
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(8)
    cb = pk.cipher_bytes()
    for extra in (8, cb, 3 * cb + 5):                  # <= 2 WT limbs, > 2 WT limbs (chunked Horner reduction)
        st = cb + extra
        big = [rng.getrandbits(8 * st) for _ in range(6)]
        k = rng.randrange(n)
        kb = np.frombuffer(k.to_bytes(128, "big"), dtype=np.uint8).copy()
        out = np.zeros((6, cb), dtype=np.uint8)
        cbuf = ints_to_be(big, st)           # (keep the array alive across the call: _ptr() only takes its address)
        rc = ctx.lib.pgpu_const_mult(pk.h, 0, 6, _ptr(cbuf), st, _ptr(kb), 128, 0, _ptr(out), cb, MEM_HOST)
        assert rc == 0
        assert be_to_ints(out) == [pow(c, k, n2) for c in big]
        assert pk.AddBatch(big, big[::-1]) == [a * b % n2 for a, b in zip(big, big[::-1])]
    # plaintexts and randomness wider than n
    ms = [rng.getrandbits(1500) for _ in range(5)]
    rs = [rng.getrandbits(1400) | 1 for _ in range(5)]
    mb, rb = ints_to_be(ms, 200), ints_to_be(rs, 180)
    out = np.zeros((5, cb), dtype=np.uint8)
    pk.encrypt_with_r_raw(5, mb, 200, rb, 180, out, cb)
    assert be_to_ints(out) == [po.encrypt_with_r(sk_o, m, r).C for m, r in zip(ms, rs)]


def test_ddleq_prover_with_a_non_unit_b(ctx):
    """The prover's response for challenge bit 1 inverts (s^an b)^en modulo n^3 (ddleq.go:107-110).  The engine computes it as
    s^(xn - an en) b^(-en) -- valid for units only: with b a multiple of p the unit test must send the batch down the literal
    sequence, whose inversion then fails as before (mpz_invert of a non-unit is undefined in the reference); instances with
    challenge bit 0 never touch that code and must still come out right."""
    import json
    import os
    import paillier_amd as pa
    from oracle import paillier_oracle as po
    G_ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    k = json.load(open(os.path.join(G_, "keys.json")))["paillier"]["2048"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    rng = random.Random(808)
    B = 6
    ct1 = [po.encrypt_with_r_at_level(sk_o, po.encrypt_with_r(sk_o, rng.randrange(n), po.rand_unit(n, rng)).C,
                                      po.rand_unit(n, rng), po.ENC_LEVEL_TWO).C for _ in range(B)]
    a_s = [po.rand_unit(n, rng) for _ in range(B)]
    b_s = [p * rng.randrange(1, q) for _ in range(B)]                   # non-units
    ct2 = [po.nested_randomize_with_ab(sk_o, po.Ciphertext(c, 1), a, b).C for c, a, b in zip(ct1, a_s, b_s)]
    xs, ys = [po.rand_unit(n, rng) for _ in range(B)], [po.rand_unit(n, rng) for _ in range(B)]
    n2, n3 = n * n, n ** 3
    bits = [po.random_oracle_bit(c1, c2, x, y, pow(c1, pow(x, n, n2), n3) * pow(y, n2, n3) % n3)
            for c1, c2, x, y in zip(ct1, ct2, xs, ys)]
    assert any(bits) and not all(bits)
    zero = [i for i, b in enumerate(bits) if not b]
    pick = lambda v: [v[i] for i in zero]
    al, es, fs = sk.ProveDDLEQInstancesBatch(pick(ct1), pick(ct2), pick(a_s), pick(b_s), pick(xs), pick(ys))
    assert (es, fs) == (pick(xs), pick(ys))
    assert pk.VerifyDDLEQInstancesBatch(pick(ct1), pick(ct2), pick(xs), pick(ys), al, es, fs) == [True] * len(zero)
    with pytest.raises(pa.PaillierHipError) as ei:
        sk.ProveDDLEQInstancesBatch(ct1, ct2, a_s, b_s, xs, ys)
    assert ei.value.code == -5            # PGPU_ERR_NOT_INVERTIBLE
