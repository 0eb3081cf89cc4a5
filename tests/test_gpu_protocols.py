"""GPU parity for the proof protocols: ExtractRandonness, NestedRandomize, the share-decryption ZKP
(thresholdkey.go:225-311) and DDLEQ (ddleq.go), against the oracle with the random draws supplied."""
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


def test_extract_randomness_both_levels(ctx):
    """operations_test.go:130-163: chosen r = i^2."""
    import paillier_amd as pa
    from paillier_amd import protocols as pr
    sk_o, p, q = po.keygen_seeded(1024, 1)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(5)
    rs = [(i * i) % n or 1 for i in range(2, 14)]
    ms = [rng.randrange(n) for _ in rs]
    for level, olevel in ((pa.ENC_LEVEL_ONE, po.ENC_LEVEL_ONE), (pa.ENC_LEVEL_TWO, po.ENC_LEVEL_TWO)):
        cts = pk.EncryptWithRBatch(ms, rs, level=level)
        got = pr.extract_randomness_batch(sk, cts, level=level)
        assert got == [po.extract_randomness(sk_o, po.Ciphertext(c, olevel)) for c in cts]
        if level == pa.ENC_LEVEL_ONE:
            assert got == rs


def test_share_decryption_zkp(ctx):
    import paillier_amd as pa
    from paillier_amd import protocols as pr
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"]["512"]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]]
    v, vks = int(k["v"], 16), [int(x, 16) for x in k["vks"]]
    tsk = po.ThresholdSecretKey(N=n, G=n + 1, TotalNumberOfDecryptionServers=total, Threshold=thr, VerificationKey=v,
                                VerificationKeys=vks, ID=3, Share=shares[2])
    rng = random.Random(9)
    cts = [po.encrypt_with_r(tsk, rng.randrange(n), po.rand_unit(n, rng)).C for _ in range(12)]
    rs = [rng.randrange(n * n) for _ in cts]
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    proofs = pr.partial_decryption_with_zkp_batch(tk, 3, shares[2], v, cts, rs)
    for pf, c, r in zip(proofs, cts, rs):
        ref = po.partial_decryption_with_zkp_r(tsk, c, r)
        assert (pf.Decryption, pf.E, pf.Z) == (ref.Decryption, ref.E, ref.Z)
    assert pr.verify_proof_batch(tk, v, vks, proofs) == [True] * len(proofs)
    # tampering (thresholdkey_test.go:283-292,385-393): wrong E, wrong ID, wrong ciphertext
    proofs[0].E += 1
    proofs[1].ID = 4
    proofs[2].C = cts[3]
    bad = pr.verify_proof_batch(tk, v, vks, proofs)
    assert bad[:3] == [False, False, False] and all(bad[3:])
    for pf, ok in zip(proofs[:4], bad[:4]):
        ref = po.PartialDecryptionZKP(ID=pf.ID, Decryption=pf.Decryption, Key=po.threshold_public_key_of(tsk), E=pf.E,
                                      Z=pf.Z, C=pf.C)
        assert po.verify_proof(ref) == ok


def test_ddleq_prove_verify(ctx):
    """ddleq_test.go:9-72 at a 256-bit key with the draws supplied: completeness, soundness, transcript parity."""
    import paillier_amd as pa
    from paillier_amd import protocols as pr
    sk_o, p, q = po.keygen_seeded(256, 3)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(21)
    B = 24
    ms = [rng.randrange(n) for _ in range(B)]
    inner = [po.encrypt_with_r(sk_o, m, po.rand_unit(n, rng)).C for m in ms]
    ct1 = [po.encrypt_with_r_at_level(sk_o, c, po.rand_unit(n, rng), po.ENC_LEVEL_TWO).C for c in inner]
    a_s = [po.rand_unit(n, rng) for _ in range(B)]
    b_s = [po.rand_unit(n, rng) for _ in range(B)]
    ct2 = pr.nested_randomize_with_ab_batch(pk, ct1, a_s, b_s)
    assert ct2 == [po.nested_randomize_with_ab(sk_o, po.Ciphertext(c, po.ENC_LEVEL_TWO), a, b).C
                   for c, a, b in zip(ct1, a_s, b_s)]
    xs = [po.rand_unit(n, rng) for _ in range(B)]
    ys = [po.rand_unit(n, rng) for _ in range(B)]
    proofs = pr.prove_ddleq_instances(sk, ct1, ct2, a_s, b_s, xs, ys)
    bits = []
    for i, pf in enumerate(proofs):
        ref = po.prove_ddleq_instance_xy(sk_o, po.Ciphertext(ct1[i], po.ENC_LEVEL_TWO), po.Ciphertext(ct2[i], po.ENC_LEVEL_TWO),
                                         a_s[i], b_s[i], xs[i], ys[i])
        assert (pf.X, pf.Y, pf.Alpha, pf.E, pf.F) == (ref.X, ref.Y, ref.Alpha, ref.E, ref.F)
        bits.append(pf.E != pf.X or pf.F != pf.Y)
    assert any(bits) and not all(bits), "both challenge bits must occur in the sample"
    assert pr.verify_ddleq_instances(pk, ct1, ct2, proofs) == [True] * B
    # soundness check of ddleq_test.go:40-72: a different ct2 must not verify
    wrong = ct2[1:] + ct2[:1]
    res = pr.verify_ddleq_instances(pk, ct1, wrong, proofs)
    want = [po.verify_ddleq_proof_instance(sk_o, po.Ciphertext(c1, po.ENC_LEVEL_TWO), po.Ciphertext(c2, po.ENC_LEVEL_TWO),
                                           po.DDLEQProofInstance(pf.X, pf.Y, pf.Alpha, pf.E, pf.F))
            for c1, c2, pf in zip(ct1, wrong, proofs)]
    assert res == want and not all(res)
    # a false statement makes the prover panic (ddleq.go:68)
    with pytest.raises(pr.DDLEQPanic):
        pr.prove_ddleq_instances(sk, ct1[:2], wrong[:2], a_s[:2], b_s[:2], xs[:2], ys[:2])


def test_random_oracle_quirks():
    from paillier_amd import protocols as pr
    assert pr.random_oracle_digest(1, 2, 3) == pr.random_oracle_digest(999, 2, 3)        # arg 0 is skipped
    assert pr.random_oracle_digest(1, 0, 3) == pr.random_oracle_digest(1, 3)             # zero contributes no bytes
    assert pr.random_oracle_digest(5, 2, 3) == po.random_oracle_digest(5, 2, 3)
    assert pr.random_oracle_bit(7, 2 ** 300 + 5, 11) == po.random_oracle_bit(7, 2 ** 300 + 5, 11)


def test_device_sha256_transcripts(ctx):
    """SHA-256 over gmp.Int.Bytes() concatenations on the device vs hashlib: every length class around the 55/56/64-byte
    padding boundaries, zero parts (empty bytes), multi-block messages."""
    import hashlib
    rng = random.Random(3)
    def by(x):
        return x.to_bytes((x.bit_length() + 7) // 8, "big")
    col0, col1, col2 = [], [], []
    for nbytes in list(range(0, 70)) + [119, 120, 127, 128, 129, 255, 256, 700]:
        a = rng.getrandbits(8 * nbytes) | (1 << (8 * nbytes - 1)) if nbytes else 0
        col0.append(a)
        col1.append(rng.choice([0, 1, 255, 256, rng.getrandbits(300)]))
        col2.append(rng.getrandbits(rng.choice([1, 8, 9, 2047, 6144])))
    got = ctx.random_oracle_digest_batch([col0, col1, col2])
    want = [hashlib.sha256(by(a) + by(b) + by(c)).digest() for a, b, c in zip(col0, col1, col2)]
    assert got == want


def test_ddleq_verify_on_device(ctx):
    """pgpu_ddleq_verify (hash + three modexps per instance on the device) vs the oracle, valid and invalid proofs."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(256, 3)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(22)
    B = 40
    ct1 = [po.encrypt_with_r_at_level(sk_o, po.encrypt_with_r(sk_o, rng.randrange(n), po.rand_unit(n, rng)).C,
                                      po.rand_unit(n, rng), po.ENC_LEVEL_TWO).C for _ in range(B)]
    a_s = [po.rand_unit(n, rng) for _ in range(B)]
    b_s = [po.rand_unit(n, rng) for _ in range(B)]
    ct2 = [po.nested_randomize_with_ab(sk_o, po.Ciphertext(c, po.ENC_LEVEL_TWO), a, b).C for c, a, b in zip(ct1, a_s, b_s)]
    proofs = [po.prove_ddleq_instance_xy(sk_o, po.Ciphertext(c1, 1), po.Ciphertext(c2, 1), a, b, po.rand_unit(n, rng),
                                         po.rand_unit(n, rng)) for c1, c2, a, b in zip(ct1, ct2, a_s, b_s)]
    args = lambda c2s, prs: (ct1, c2s, [p.X for p in prs], [p.Y for p in prs], [p.Alpha for p in prs], [p.E for p in prs],
                             [p.F for p in prs])
    assert pk.VerifyDDLEQInstancesBatch(*args(ct2, proofs)) == [True] * B
    wrong = ct2[1:] + ct2[:1]
    got = pk.VerifyDDLEQInstancesBatch(*args(wrong, proofs))
    want = [po.verify_ddleq_proof_instance(sk_o, po.Ciphertext(c1, 1), po.Ciphertext(c2, 1), pf)
            for c1, c2, pf in zip(ct1, wrong, proofs)]
    assert got == want and not all(got)
    proofs[0].F += 1
    proofs[1].E = (proofs[1].E + 1) % (n * n)
    got = pk.VerifyDDLEQInstancesBatch(*args(ct2, proofs))
    assert got[:2] == [False, False] and all(got[2:])


@pytest.mark.parametrize("bits", ["512", "2048"])
def test_share_zkp_device_resident(ctx, bits):
    """pgpu_share_zkp_prove / _verify (thresholdkey.go:225-311) fully on the device -- comb table for V^r, device SHA-256 over
    the unreduced c^4 and c_i^2 -- against the oracle: same (Decryption, E, Z), acceptance, and rejection of tampering."""
    import paillier_amd as pa
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"][bits]
    n, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]]
    v, vks = int(k["v"], 16), [int(x, 16) for x in k["vks"]]
    sid = 2
    tsk = po.ThresholdSecretKey(N=n, G=n + 1, TotalNumberOfDecryptionServers=total, Threshold=thr, VerificationKey=v,
                                VerificationKeys=vks, ID=sid, Share=shares[sid - 1])
    rng = random.Random(int(bits) + 5)
    B = 10
    cts = [po.encrypt_with_r(tsk, rng.randrange(n), po.rand_unit(n, rng)).C for _ in range(B)]
    rs = [rng.randrange(n * n) for _ in cts]
    rs[0] = 0
    tk = pa.ThresholdPublicKey(ctx, n, total=total, threshold=thr)
    dec, es, zs = tk.PartialDecryptionWithZKPBatch(sid, shares[sid - 1], v, cts, rs)
    refs = [po.partial_decryption_with_zkp_r(tsk, c, r) for c, r in zip(cts, rs)]
    assert dec == [p.Decryption for p in refs]
    assert es == [p.E for p in refs]
    assert zs == [p.Z for p in refs]
    assert tk.VerifyProofBatch(v, vks[sid - 1], cts, dec, es, zs) == [True] * B
    es2, zs2, cts2 = list(es), list(zs), list(cts)
    es2[0] ^= 1
    zs2[1] += 1
    cts2[2] = cts[3]
    got = tk.VerifyProofBatch(v, vks[sid - 1], cts2, dec, es2, zs2)
    assert got[:3] == [False, False, False] and all(got[3:])
    # wrong verification key (another server's): thresholdkey_test.go:294-327 expects rejection
    assert tk.VerifyProofBatch(v, vks[sid % total], cts, dec, es, zs) == [False] * B
    # a verification key v_i that is not a unit modulo n^2 has no inverse (thresholdkey.go:308: mpz_invert undefined): every proof
    # under it is rejected; and the engine's inverse of v_i (taken once, kept with the key) serves a second call
    p_ = int(k["p"], 16) if "p" in k else None
    if p_:
        assert tk.VerifyProofBatch(v, p_ * 12345, cts, dec, es, zs) == [False] * B
    assert tk.VerifyProofBatch(v, vks[sid - 1], cts, dec, es, zs) == [True] * B


def test_ddleq_prove_on_device(ctx):
    """pgpu_ddleq_prove (sanity check, Alpha, device Fiat-Shamir bit, level-two ExtractRandonness response) vs the oracle."""
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(256, 3)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(23)
    B = 30
    ct1 = [po.encrypt_with_r_at_level(sk_o, po.encrypt_with_r(sk_o, rng.randrange(n), po.rand_unit(n, rng)).C,
                                      po.rand_unit(n, rng), po.ENC_LEVEL_TWO).C for _ in range(B)]
    a_s = [po.rand_unit(n, rng) for _ in range(B)]
    b_s = [po.rand_unit(n, rng) for _ in range(B)]
    ct2 = [po.nested_randomize_with_ab(sk_o, po.Ciphertext(c, 1), a, b).C for c, a, b in zip(ct1, a_s, b_s)]
    xs = [po.rand_unit(n, rng) for _ in range(B)]
    ys = [po.rand_unit(n, rng) for _ in range(B)]
    al, es, fs = sk.ProveDDLEQInstancesBatch(ct1, ct2, a_s, b_s, xs, ys)
    refs = [po.prove_ddleq_instance_xy(sk_o, po.Ciphertext(c1, 1), po.Ciphertext(c2, 1), a, b, x, y)
            for c1, c2, a, b, x, y in zip(ct1, ct2, a_s, b_s, xs, ys)]
    assert al == [r.Alpha for r in refs]
    assert es == [r.E for r in refs]
    assert fs == [r.F for r in refs]
    bits = [r.E != r.X for r in refs]
    assert any(bits) and not all(bits)
    assert pk.VerifyDDLEQInstancesBatch(ct1, ct2, xs, ys, al, es, fs) == [True] * B
    with pytest.raises(pa.PaillierHipError) as ei:     # ddleq.go:68 panics on a false statement
        sk.ProveDDLEQInstancesBatch(ct1[:3], ct2[1:4], a_s[:3], b_s[:3], xs[:3], ys[:3])
    assert "cannot prove re-encryption" in str(ei.value)


def test_ddleq_prove_3072_bit_key(ctx):
    """The prover at 3072 bits: halves modulo p^3, q^3 on the three-digit kernel for 55-limb primes (vm_asm_55_48), split by the
    base-p digits of the exponents through p^2, q^2 on the one-lane pair kernel for 55-limb primes (vm_asm_55_16, 4-bit
    per-number windows), a^n | x^n through p^2, q^2 on it as well; the verifier's n^3 is 9 216 bits (vm_asm_42_8, eight lanes per
    number: no launch falls back to the compiler-generated kernel).  Few instances: the oracle's 9 216-bit powers are slow."""
    import json as _json
    import paillier_amd as pa
    k = _json.load(open(os.path.join(G, "keys.json")))["paillier"]["3072"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    sk_o = po.SecretKey(N=n, G=n + 1, Lambda=lam)
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, lam)
    rng = random.Random(3073)      # (a seed for which both challenge bits occur among the four instances)
    B = 4
    ct1 = [po.encrypt_with_r_at_level(sk_o, po.encrypt_with_r(sk_o, rng.randrange(n), po.rand_unit(n, rng)).C,
                                      po.rand_unit(n, rng), po.ENC_LEVEL_TWO).C for _ in range(B)]
    a_s, b_s, xs, ys = ([po.rand_unit(n, rng) for _ in range(B)] for _ in range(4))
    ct2 = [po.nested_randomize_with_ab(sk_o, po.Ciphertext(c, 1), a, b).C for c, a, b in zip(ct1, a_s, b_s)]
    # lanes_wanted = 1: the kernel shapes of a chip-filling batch -- the one-lane pair kernel for 55-limb primes, and the
    # prover's p-adic split is attempted (it needs per-number windows modulo p^2, which that kernel does not have: the
    # library must fall back to the unsplit ladders, not compute garbage)
    ctx.set_flag("lanes_wanted", 1)
    try:
        al, es, fs = sk.ProveDDLEQInstancesBatch(ct1, ct2, a_s, b_s, xs, ys)
    finally:
        ctx.set_flag("lanes_wanted", 0)
    refs = [po.prove_ddleq_instance_xy(sk_o, po.Ciphertext(c1, 1), po.Ciphertext(c2, 1), a, b, x, y)
            for c1, c2, a, b, x, y in zip(ct1, ct2, a_s, b_s, xs, ys)]
    assert (al, es, fs) == ([r.Alpha for r in refs], [r.E for r in refs], [r.F for r in refs])
    assert len({r.E != r.X for r in refs}) == 2, "both challenge bits should occur (change the seed)"
    # one verifier call: the three true statements and the same proofs against a neighbour's ct2
    wrong = ct2[1:] + ct2[:1]
    got = pk.VerifyDDLEQInstancesBatch(ct1 + ct1, ct2 + wrong, xs + xs, ys + ys, al + al, es + es, fs + fs)
    assert ctx.last_vm_launches() > 0 and ctx.last_vm_asm() == ctx.last_vm_launches()      # the 9 216-bit n^3 included
    assert got[:B] == [True] * B
    assert got[B:] == [po.verify_ddleq_proof_instance(sk_o, po.Ciphertext(c1, 1), po.Ciphertext(c2, 1),
                                                      po.DDLEQProofInstance(x, y, a_, e, f))
                       for c1, c2, x, y, a_, e, f in zip(ct1, wrong, xs, ys, al, es, fs)]


def test_whole_protocol_forms(ctx):
    """ProveDDLEQ / VerifyDDLEQProof with host-drawn randomness (ddleq_test.go:9-72) and CombinePartialDecryptionsZKP /
    VerifyDecryption (thresholdkey_test.go:294-394): completeness, and a cheating server being dropped."""
    import paillier_amd as pa
    from paillier_amd import protocols as pr
    # DDLEQ, secpar = 10 as in the reference's tests
    sk_o, p, q = po.keygen_seeded(256, 3)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(41)
    ct1 = pk.NestedEncryptBatch([1234])[0]
    a, b = po.rand_unit(n, rng), po.rand_unit(n, rng)
    ct2 = pr.nested_randomize_with_ab_batch(pk, [ct1], [a], [b])[0]
    proof = pr.prove_ddleq(sk, 10, ct1, ct2, a, b)
    assert pr.verify_ddleq_proof(pk, ct1, ct2, proof)
    assert po.verify_ddleq_proof(sk_o, po.Ciphertext(ct1, 1), po.Ciphertext(ct2, 1),
                                 [po.DDLEQProofInstance(i.X, i.Y, i.Alpha, i.E, i.F) for i in proof])
    other = pr.nested_randomize_with_ab_batch(pk, [pk.NestedEncryptBatch([1235])[0]], [a], [b])[0]
    assert not pr.verify_ddleq_proof(pk, ct1, other, proof)
    # threshold with proofs
    k = json.load(open(os.path.join(G, "keys.json")))["threshold"]["512"]
    tn, total, thr = int(k["n"], 16), k["total"], k["threshold"]
    shares = [int(s, 16) for s in k["shares"]]
    v, vks = int(k["v"], 16), [int(x, 16) for x in k["vks"]]
    tk = pa.ThresholdPublicKey(ctx, tn, total=total, threshold=thr)
    ms = [rng.randrange(tn) for _ in range(6)]
    cts = tk.EncryptBatch(ms)
    def proofs_of(sid):
        rs = [rng.randrange(tn * tn) for _ in cts]
        d, es, zs = tk.PartialDecryptionWithZKPBatch(sid, shares[sid - 1], v, cts, rs)
        return [pr.PartialDecryptionZKP(sid, di, ei, zi, ci) for di, ei, zi, ci in zip(d, es, zs, cts)]
    # VerifyPartialDecryption (thresholdkey.go:258-275): a right share passes, a wrong one is an "Invalid share"
    pr.verify_partial_decryption(tk, 2, shares[1], v, vks, trials=3)
    with pytest.raises(ValueError, match="Invalid share"):
        pr.verify_partial_decryption(tk, 2, shares[1] + 1, v, vks)
    srv = [proofs_of(s) for s in (1, 2, 4, 5)]
    assert pr.combine_partial_decryptions_zkp(tk, v, vks, srv) == ms
    pr.verify_decryption(tk, v, vks, cts, ms, srv)
    srv[1][3].Decryption ^= 2          # server 2 cheats on ciphertext 3: dropped there, the other three still reach t = 3
    assert pr.combine_partial_decryptions_zkp(tk, v, vks, srv) == ms
    srv[0][3].Z += 1                   # a second bad proof on the same ciphertext: only 2 < t shares survive
    with pytest.raises(pa.PaillierHipError) as ei:
        pr.combine_partial_decryptions_zkp(tk, v, vks, srv)
    assert ei.value.code == -6
    with pytest.raises(ValueError):
        pr.verify_decryption(tk, v, vks, cts[::-1], ms, [proofs_of(1), proofs_of(2), proofs_of(3)])
