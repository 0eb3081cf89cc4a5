"""Cross-checks the two CPU oracles: the Python-int restatement and the libgmp restatement (the library the
reference itself calls through ncw/gmp).  Two independent big-integer implementations agreeing on seeded
inputs at cryptographic sizes is the pin the reference's own (toy-sized) KATs cannot give."""
import random

import pytest

from oracle import gmp_oracle as go
from oracle import paillier_oracle as po


@pytest.mark.parametrize("bits", [1024, 2048])
def test_encrypt_decrypt_cross(bits):
    sk, p, q = po.keygen_seeded(bits, bits)
    rng = random.Random(bits)
    ms = [rng.randrange(sk.N) for _ in range(6)] + [0, 1, sk.N - 1]
    rs = [po.rand_unit(sk.N, rng) for _ in ms]
    cts_py = [po.encrypt_with_r(sk, m, r).C for m, r in zip(ms, rs)]
    assert go.encrypt_batch(sk.N, sk.G, ms, rs, threads=2) == cts_py
    assert go.decrypt_batch(sk.N, sk.Lambda, cts_py, threads=2) == ms
    assert [po.decrypt(sk, po.Ciphertext(c)) for c in cts_py] == ms
    # arbitrary (not necessarily valid) ciphertexts, including non-units and zero
    weird = [rng.randrange(sk.N ** 2) for _ in range(4)] + [0, p, q * 5, sk.N, sk.N ** 2 - 1]
    assert go.decrypt_batch(sk.N, sk.Lambda, weird) == [po.decrypt(sk, po.Ciphertext(c)) for c in weird]


def test_modexp_cross():
    rng = random.Random(9)
    n = rng.getrandbits(2048) | 1 | (1 << 2047)
    bases = [rng.randrange(n) for _ in range(5)] + [0, 1]
    for e in (0, 1, rng.getrandbits(700)):
        assert go.modexp_batch(n, e, bases) == [po.gmp_exp(b, e, n) for b in bases]


def test_key4096_vectors_against_libgmp():
    """tests/golden/key4096.json was written by the Python oracle; libgmp (the reference's own backend) must reproduce its
    ciphertexts and plaintexts for the units (for non-units libgmp's mpz_invert-free path is the same formula: compare too)."""
    import json
    import os
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "key4096.json")))
    n, lam = int(k["n"], 16), int(k["lambda"], 16)
    assert n == int(k["p"], 16) * int(k["q"], 16) and n.bit_length() == 4096
    ms, rs, cs = ([int(x, 16) for x in k[key]] for key in ("m", "r", "c"))
    assert go.encrypt_batch(n, n + 1, ms, rs) == cs
    assert go.decrypt_batch(n, lam, cs) == ms
    wc, wm = ([int(x, 16) for x in k[key]] for key in ("weird_c", "weird_m"))
    assert go.decrypt_batch(n, lam, wc) == wm


def _be_rows(vals, stride):
    import numpy as np
    return np.frombuffer(b"".join(int(v).to_bytes(stride, "big") for v in vals), dtype=np.uint8).reshape(len(vals), stride).copy()


def _ints(arr):
    return [int.from_bytes(bytes(r), "big") for r in arr.reshape(-1, arr.shape[-1])]


def test_threshold_restatement_reference_kats():
    """The libgmp PartialDecrypt + Combine restatement (bench.py's CPU baseline beside BASELINE config 4) on the reference's own
    toy KATs: TestDecrypt (thresholdkey_test.go:58-74: N = 10403, share 862, l = 10, c = 56 -> partial decryption 40644522) and
    TestDecryption (:267-281: N = 637753, l = 2, shares of servers 1, 2 -> message 100 through Combine)."""
    c = _be_rows([56], 8)
    out, used, parts = go.threshold_decrypt_batch_raw(10403, 10, [1], [862], c, 8, want_partials=True)
    assert _ints(parts) == [40644522]
    # TestDecryption combines two GIVEN partial decryptions; the restatement computes its own from shares, so re-derive here: a
    # (2, 2) sharing of a toy key and every quantity through the Python-int oracle as well
    p, p1, q, q1 = 839, 419, 887, 443                     # TestInitShortcuts' safe primes (thresholdkey_generator_test.go:213-230)
    n = p * q
    assert n == 744193
    tsks = po.threshold_keys_from_primes(p, p1, q, q1, 5, 3, random.Random(7))
    rng = random.Random(8)
    cts = [po.encrypt_with_r(tsks[0], m, po.rand_unit(n, rng)).C for m in (0, 1, 100, n - 1, 31337)]
    for ids in ([1, 2, 3], [1, 3, 5], [5, 2, 4], [2, 4, 5, 1]):
        shares = [tsks[i - 1].Share for i in ids]
        out, used, parts = go.threshold_decrypt_batch_raw(n, 5, ids, shares, _be_rows(cts, 8), 4, threads=2, want_partials=True)
        want_parts = [po.partial_decrypt(tsks[i - 1], c).Decryption for c in cts for i in ids]
        assert _ints(parts) == want_parts
        assert _ints(out) == [po.combine_partial_decryptions(tsks[0], [po.partial_decrypt(tsks[i - 1], c) for i in ids]) for c in cts]
        assert _ints(out) == [0, 1, 100, n - 1, 31337]


def test_threshold_restatement_against_the_2048_bit_fixtures():
    """... and on the committed 2048-bit fixture (tests/golden/proofs.json: partials of all five servers, Combine for every
    3-subset of 5 -- every subset has a negative Lagrange coefficient, i.e. the ModInverse branch of thresholdkey.go:134-138)."""
    import itertools
    import json
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    t = json.load(open(os.path.join(G, "keys.json")))["threshold"]["2048"]
    th = json.load(open(os.path.join(G, "proofs.json")))["threshold"]
    n, shares = int(t["n"], 16), [int(s, 16) for s in t["shares"]]
    cs, ms = [int(x, 16) for x in th["c"]], [int(x, 16) for x in th["m"]]
    for ids in list(itertools.combinations(range(1, 6), 3))[:4] + [(5, 3, 1)]:
        out, used, parts = go.threshold_decrypt_batch_raw(n, 5, list(ids), [shares[i - 1] for i in ids], _be_rows(cs, 512), 256,
                                                          threads=2, want_partials=True)
        assert _ints(out) == ms
        got = parts.reshape(len(cs), 3, 512)
        for k, i in enumerate(ids):
            assert _ints(got[:, k, :]) == [int(x, 16) for x in th["partials"][i - 1]]


def test_level_two_and_nested_randomize_restatements():
    """libgmp EncryptWithRAtLevel (level two) and NestedRandomize against the Python-int oracle and the committed fixtures."""
    import json
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    k = json.load(open(os.path.join(G, "keys.json")))["paillier"]["2048"]
    P = json.load(open(os.path.join(G, "proofs.json")))
    n = int(k["n"], 16)
    l2 = P["level2"]
    ms, rs, cs = ([int(x, 16) for x in l2[key]] for key in ("m", "r", "c"))
    rstride = max(256, max((r.bit_length() + 7) // 8 for r in rs))
    out, _ = go.encrypt_l2_batch_raw(n, n + 1, _be_rows(ms, 512), _be_rows(rs, rstride), 768, threads=2)
    assert _ints(out) == cs
    st = P["ddleq"]["statements"]
    ct1, ct2, a, b = ([int(s[key], 16) for s in st] for key in ("ct1", "ct2", "a", "b"))
    out, _ = go.nested_randomize_batch_raw(n, _be_rows(ct1, 768), _be_rows(a, 256), _be_rows(b, 256), threads=2)
    assert _ints(out) == ct2
    pk = po.PublicKey(N=n, G=n + 1)
    assert ct2[:2] == [po.nested_randomize_with_ab(pk, po.Ciphertext(c, 1), x, y).C for c, x, y in zip(ct1[:2], a[:2], b[:2])]


def _golden():
    import json
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return (json.load(open(os.path.join(G, "keys.json"))), json.load(open(os.path.join(G, "vectors.json"))),
            json.load(open(os.path.join(G, "proofs.json"))))


def test_add_sub_const_mult_restatements():
    """libgmp Add / Sub / ConstMult (bench.py's CPU figures beside add_2048 / sub_2048 / const_mult_2048) against the committed
    fixtures and the Python-int oracle, both levels; a subtrahend without an inverse is flagged."""
    K, V, P = _golden()
    k = K["paillier"]["2048"]
    n = int(k["n"], 16)
    n2, n3 = n * n, n ** 3
    pk = po.PublicKey(N=n, G=n + 1)
    v = V["2048"]
    a, b, want = ([int(x, 16) for x in v["add"][key]] for key in ("a", "b", "out"))
    out, _, ok = go.add_sub_batch_raw(n2, False, _be_rows(a, 512), _be_rows(b, 512), 512, threads=2)
    assert _ints(out) == want == [po.add(pk, po.Ciphertext(x), po.Ciphertext(y)).C for x, y in zip(a, b)] and ok.all()
    out, _, ok = go.add_sub_batch_raw(n2, True, _be_rows(a, 512), _be_rows(b, 512), 512, threads=2)
    assert _ints(out) == [po.sub(pk, po.Ciphertext(x), po.Ciphertext(y)).C for x, y in zip(a, b)] and ok.all()
    rng = random.Random(31)
    a3, b3 = [rng.randrange(n3) for _ in range(5)], [rng.randrange(n3) | 1 for _ in range(5)]
    import math
    b3 = [x for x in b3 if math.gcd(x, n) == 1]
    a3 = a3[:len(b3)]
    out, _, ok = go.add_sub_batch_raw(n3, True, _be_rows(a3, 768), _be_rows(b3, 768), 768)
    assert _ints(out) == [po.sub(pk, po.Ciphertext(x, 1), po.Ciphertext(y, 1)).C for x, y in zip(a3, b3)] and ok.all()
    p_ = int(k["p"], 16)
    out, _, ok = go.add_sub_batch_raw(n2, True, _be_rows(a[:2], 512), _be_rows([p_ * 7, b[1]], 512), 512)
    assert list(ok) == [0, 1] and _ints(out)[0] == 0
    # ConstMult: a shared k (the fixture's, and BenchmarkConstMul2's 50^50 mod n^2) and one k per ciphertext
    cs, ks = ([int(x, 16) for x in v["const_mult"][key]] for key in ("c", "k"))
    out, _ = go.const_mult_batch_raw(n2, _be_rows(cs, 512), ks[0], 512, threads=2)
    assert _ints(out) == [int(x, 16) for x in v["const_mult"]["out_shared_k0"]]
    kl = max(1, max((x.bit_length() + 7) // 8 for x in ks))
    out, _ = go.const_mult_batch_raw(n2, _be_rows(cs, 512), _be_rows(ks, kl), 512, threads=2)
    assert _ints(out) == [int(x, 16) for x in v["const_mult"]["out"]]
    k50 = pow(50, 50, n2)
    out, _ = go.const_mult_batch_raw(n2, _be_rows(cs[:3], 512), k50, 512)
    assert _ints(out) == [po.const_mult(pk, po.Ciphertext(c), k50).C for c in cs[:3]]
    out, _ = go.const_mult_batch_raw(n2, _be_rows(cs[:2], 512), 0, 512)             # gmp.Int.Exp(x, 0) = 1
    assert _ints(out) == [1, 1]


def test_alt_encrypt_restatement():
    K, V, P = _golden()
    k = K["paillier"]["2048"]
    n, h, kk = int(k["n"], 16), int(k["h"], 16), int(k["k"], 16)
    pk = po.PublicKey(N=n, G=n + 1, H=h, K=kk)
    rng = random.Random(77)
    ms = [0, 1, n - 1] + [rng.randrange(n) for _ in range(4)]
    rs = [0, kk, kk + 5] + [rng.randrange(n) for _ in range(4)]
    out, _, rred = go.alt_encrypt_batch_raw(n, n + 1, h, kk, _be_rows(ms, 256), _be_rows(rs, 256), 512, threads=2)
    want = [po.alt_encrypt_with_r_at_level(pk, m, r, 0) for m, r in zip(ms, rs)]
    assert _ints(out) == [w[0].C for w in want]
    assert _ints(rred) == [w[1] for w in want] == [r % kk for r in rs]


def test_share_zkp_restatement_fixtures_and_reference_kats():
    """libgmp PartialDecryptionWithZKP / VerifyProof: the reference's own KATs for verifyPart1 / verifyPart2
    (thresholdkey_test.go:109-135: 11986, 14602), the committed 2048-bit proofs (hash transcript included) and the Python oracle."""
    # TestVerifyPart1: N = 131, C = 99, Decryption = 101, E = 112, Z = 88 -> a = 11986; TestVerifyPart2: V = 101, v_i = 77 -> b = 14602
    e112 = (112).to_bytes(32, "big")
    import numpy as np
    ok, _, ab = go.share_zkp_verify_batch_raw(131, 101, 77, _be_rows([99], 4), _be_rows([101], 4),
                                              np.frombuffer(e112, dtype=np.uint8).reshape(1, 32).copy(), _be_rows([88], 4), want_ab=True)
    assert _ints(ab) == [11986, 14602] and list(ok) == [0]
    K, V, P = _golden()
    t = K["threshold"]["2048"]
    n, v, vks, shares = int(t["n"], 16), int(t["v"], 16), [int(x, 16) for x in t["vks"]], [int(x, 16) for x in t["shares"]]
    server = P["share_zkp"]["server"]
    recs = P["share_zkp"]["proofs"]
    col = lambda key: [int(r[key], 16) for r in recs]
    cs, rs = col("c"), col("r")
    zs_len = 512 + 48
    dec, eo, zo, _ = go.share_zkp_prove_batch_raw(n, int(t["total"]), shares[server - 1], v, _be_rows(cs, 512), _be_rows(rs, 512), zs_len,
                                                  threads=2)
    assert _ints(dec) == col("dec") and _ints(eo) == col("e") and _ints(zo) == col("z")
    ok, _, ab = go.share_zkp_verify_batch_raw(n, v, vks[server - 1], _be_rows(cs, 512), dec, eo, zo, threads=2, want_ab=True)
    assert ok.all()
    assert [x for x in _ints(ab)[0::2]] == col("verify_a") and [x for x in _ints(ab)[1::2]] == col("verify_b")
    # a tampered Z, a tampered E, another server's verification key: rejected
    zbad = zo.copy(); zbad[0, -1] ^= 1
    ebad = eo.copy(); ebad[1, 0] ^= 0x80
    assert list(go.share_zkp_verify_batch_raw(n, v, vks[server - 1], _be_rows(cs, 512), dec, eo, zbad)[0])[:2] == [0, 1]
    assert list(go.share_zkp_verify_batch_raw(n, v, vks[server - 1], _be_rows(cs, 512), dec, ebad, zo)[0])[:2] == [1, 0]
    assert not go.share_zkp_verify_batch_raw(n, v, vks[server % 5], _be_rows(cs, 512), dec, eo, zo)[0].any()
    # and against the Python-int restatement on a fresh draw
    tsk = po.ThresholdSecretKey(N=n, G=n + 1, TotalNumberOfDecryptionServers=int(t["total"]), Threshold=int(t["threshold"]),
                                VerificationKey=v, VerificationKeys=vks, ID=server, Share=shares[server - 1])
    rng = random.Random(3)
    c1, r1 = rng.randrange(n * n), rng.randrange(n * n)
    pd = po.partial_decryption_with_zkp_r(tsk, c1, r1)
    dec, eo, zo, _ = go.share_zkp_prove_batch_raw(n, int(t["total"]), shares[server - 1], v, _be_rows([c1], 512), _be_rows([r1], 512), zs_len)
    assert (_ints(dec), _ints(eo), _ints(zo)) == ([pd.Decryption], [pd.E], [pd.Z]) and po.verify_proof(pd)


def test_level_two_decrypt_restatement():
    """libgmp Decrypt at level two (bench.py's CPU figure beside decrypt_l2_2048) against the Python-int oracle and the committed
    level-two fixtures, arbitrary elements of Z_{n^3} included (the reference's recovery algorithm on non-ciphertexts)."""
    K, V, P = _golden()
    k = K["paillier"]["2048"]
    n, lam = int(k["n"], 16), int(k["lambda"], 16)
    sk = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    l2 = P["level2"]
    ms, cs = ([int(x, 16) for x in l2[key]] for key in ("m", "c"))
    out, _ = go.decrypt_l2_batch_raw(n, lam, _be_rows(cs, 768), 512, threads=2)
    assert _ints(out) == [m % (n * n) for m in ms]
    rng = random.Random(12)
    weird = [rng.randrange(n ** 3) for _ in range(3)] + [1, n ** 3 - 1]
    out, _ = go.decrypt_l2_batch_raw(n, lam, _be_rows(weird, 768), 512)
    assert _ints(out) == [po.decrypt(sk, po.Ciphertext(c, po.ENC_LEVEL_TWO)) for c in weird]
