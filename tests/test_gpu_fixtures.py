"""GPU parity against the committed fixtures of the proof / threshold / level-two paths at the BASELINE key size
(tests/golden/proofs.json: 2048-bit keys, written by make_golden.py from the Python-int oracle and re-derived by the
libgmp oracle in tests/test_golden_proofs.py).  Nothing here is computed by an oracle at test time: the HIP path, through
the C ABI, against committed numbers.

BASELINE config 5 (DDLEQ, 2048 bits) runs at its real size: 64 (statement, instance) pairs, so n^3 is 6144 bits and the
prover takes pow_n3_crt over p^3 / q^3 with the interleaved ladders, the verifier the 6144-bit interleaved ladder."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def H(xs):
    return [int(x, 16) for x in xs]


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


@pytest.fixture(scope="module")
def keys(ctx):
    import paillier_amd as pa
    k = load("keys.json")["paillier"]["2048"]
    n, lam = int(k["n"], 16), int(k["lambda"], 16)
    pk = pa.PublicKey(ctx, n, n + 1)
    return pk, pa.SecretKey(ctx, pk, lam)


def test_ddleq_2048_prove_and_verify(ctx, keys):
    """pgpu_ddleq_prove / pgpu_ddleq_verify (ddleq.go:55-153) on 64 instances of 4 statements, both challenge bits."""
    pk, sk = keys
    d = load("proofs.json")["ddleq"]
    st = [{k: int(v, 16) for k, v in s.items()} for s in d["statements"]]
    ins = d["instances"]
    col = lambda key: [st[i["s"]][key] for i in ins]
    xs, ys = H(i["x"] for i in ins), H(i["y"] for i in ins)
    al, es, fs = sk.ProveDDLEQInstancesBatch(col("ct1"), col("ct2"), col("a"), col("b"), xs, ys)
    assert ctx.last_vm_asm() > 0
    dg = [hashlib.sha256(a.to_bytes(768, "big") + e.to_bytes(512, "big") + f.to_bytes(768, "big")).hexdigest()
          for a, e, f in zip(al, es, fs)]
    assert dg == [i["digest"] for i in ins]
    for i, rec in enumerate(ins[:16]):
        assert (al[i], es[i], fs[i]) == (int(rec["alpha"], 16), int(rec["e"], 16), int(rec["f"], 16))
    bits = [int(e != x or f != y) for e, f, x, y in zip(es, fs, xs, ys)]
    assert bits == [i["bit"] for i in ins] and 0 < sum(bits) < 64
    assert pk.VerifyDDLEQInstancesBatch(col("ct1"), col("ct2"), xs, ys, al, es, fs) == [True] * 64
    wrong = [st[(i["s"] + 1) % 4]["ct2"] for i in ins]
    got = pk.VerifyDDLEQInstancesBatch(col("ct1"), wrong, xs, ys, al, es, fs)
    assert got == [bool(i["verify_wrong_ct2"]) for i in ins] and not all(got)
    # tampered responses are rejected
    es2, fs2 = list(es), list(fs)
    es2[0] ^= 1
    fs2[1] ^= 1
    got = pk.VerifyDDLEQInstancesBatch(col("ct1"), col("ct2"), xs, ys, al, es2, fs2)
    assert got[:2] == [False, False] and all(got[2:])
    # a false statement: the reference panics (ddleq.go:68)
    import paillier_amd as pa
    with pytest.raises(pa.PaillierHipError) as ei:
        sk.ProveDDLEQInstancesBatch(col("ct1")[:4], wrong[:4], col("a")[:4], col("b")[:4], xs[:4], ys[:4])
    assert "cannot prove re-encryption" in str(ei.value)


def _digests(al, es, fs):
    return [hashlib.sha256(a.to_bytes(768, "big") + e.to_bytes(512, "big") + f.to_bytes(768, "big")).hexdigest()
            for a, e, f in zip(al, es, fs)]


def test_ddleq_2048_secpar40_hoisted(ctx, keys):
    """pgpu_ddleq_prove_secpar (ProveDDLEQ, ddleq.go:27-40, at the reference's test setting secpar = 40: ddleq_test.go:74-88) for
    two statements: the sanity check, a^n, a^-1 and ExtractRandonness(ct1) are computed once per statement, and every Alpha / E /
    F must be byte for byte what the oracle's proveDDLEQInstance gave for that instance on its own (ddleq_secpar40.json)."""
    import paillier_amd as pa
    from paillier_amd import protocols as pr
    pk, sk = keys
    d = load("ddleq_secpar40.json")
    st = [{k: int(v, 16) for k, v in s.items()} for s in load("proofs.json")["ddleq"]["statements"]]
    ss = [st[p["statement"]] for p in d["proofs"]]
    xs = [H(i["x"] for i in p["instances"]) for p in d["proofs"]]
    ys = [H(i["y"] for i in p["instances"]) for p in d["proofs"]]
    col = lambda key: [s[key] for s in ss]
    al, es, fs = sk.ProveDDLEQBatch(40, col("ct1"), col("ct2"), col("a"), col("b"), xs, ys)
    assert ctx.last_vm_asm() > 0
    for j, p in enumerate(d["proofs"]):
        ins = p["instances"]
        assert _digests(al[j], es[j], fs[j]) == [i["digest"] for i in ins], j
        for k, rec in enumerate(ins[:4]):
            assert (al[j][k], es[j][k], fs[j][k]) == (int(rec["alpha"], 16), int(rec["e"], 16), int(rec["f"], 16))
        bits = [int(e != x or f != y) for e, f, x, y in zip(es[j], fs[j], xs[j], ys[j])]
        assert bits == [i["bit"] for i in ins] and 0 < sum(bits) < 40
    # one statement alone (n_statements = 1), and the secpar = 1 entry point on the same rows: the same integers
    a1, e1, f1 = sk.ProveDDLEQBatch(40, col("ct1")[1:], col("ct2")[1:], col("a")[1:], col("b")[1:], xs[1:], ys[1:])
    assert (a1[0], e1[0], f1[0]) == (al[1], es[1], fs[1])
    a2, e2, f2 = sk.ProveDDLEQInstancesBatch([ss[0]["ct1"]] * 40, [ss[0]["ct2"]] * 40, [ss[0]["a"]] * 40, [ss[0]["b"]] * 40, xs[0], ys[0])
    assert (a2, e2, f2) == (al[0], es[0], fs[0])
    # VerifyDDLEQProof per statement (ddleq.go:44-53): all instances of a statement must verify
    proofs = [[pr.DDLEQProofInstance(xs[j][k], ys[j][k], al[j][k], es[j][k], fs[j][k]) for k in range(40)] for j in range(2)]
    assert pr.verify_ddleq_proof_batch(pk, col("ct1"), col("ct2"), proofs) == [True, True]
    proofs[1][17] = pr.DDLEQProofInstance(xs[1][17], ys[1][17], al[1][17], es[1][17], fs[1][17] ^ 1)
    assert pr.verify_ddleq_proof_batch(pk, col("ct1"), col("ct2"), proofs) == [True, False]
    # a false statement among true ones: the reference panics (ddleq.go:68)
    with pytest.raises(pa.PaillierHipError) as ei:
        sk.ProveDDLEQBatch(40, col("ct1"), [ss[0]["ct2"], ss[0]["ct2"]], col("a"), col("b"), xs, ys)
    assert "cannot prove re-encryption" in str(ei.value)


def test_ddleq_2048_secpar16_from_the_64_instance_fixture(ctx, keys):
    """The 64 committed instances of proofs.json are 16 instances of each of 4 statements: as ONE pgpu_ddleq_prove_secpar call
    (4 statements, secpar 16; also with the p-adic split forced) they must land on the same digests."""
    pk, sk = keys
    d = load("proofs.json")["ddleq"]
    st = [{k: int(v, 16) for k, v in s.items()} for s in d["statements"]]
    ins = d["instances"]
    by = [[i for i in ins if i["s"] == j] for j in range(4)]
    xs, ys = [H(i["x"] for i in b) for b in by], [H(i["y"] for i in b) for b in by]
    col = lambda key: [s[key] for s in st]
    # side = 0: the per-statement chains on the one stream, in order.  lanes_wanted = 1: the one-lane pair kernel, whose per-number
    # window tables are number-major (VM_STORET / VM_MULVT); nm4 = 0: the same ladders on limb-major tables (VM_MULV)
    # early = 0: the response is prepared after the hash for the instances with challenge bit 1 (default: for every statement and
    # instance beside the Alpha ladders, gathered afterwards)
    # struct = 0: the sanity values and Alpha by ladders on ct1 itself (pow_n3_crt) instead of through the structure of the unit
    # group (struct_pow_n3: plaintext of ct1, ladders modulo the primes, Teichmueller lift)
    # exclusive = 0: the placement of the call's concurrent launches left to the dispatcher (default for a call this small: every
    # workgroup asks for a whole compute unit's LDS; with lanes_wanted = 1 the plan never asks for it)
    for lanes_wanted, side, nm4, early, struct, excl in ((0, 1, 1, 1, 1, 1), (0, 1, 1, 1, 1, 0), (1, 1, 1, 1, 1, 1), (1, 1, 0, 1, 1, 1),
                                                          (0, 0, 1, 1, 1, 1), (0, 1, 1, 0, 1, 1), (1, 0, 1, 0, 1, 1), (0, 1, 1, 1, 0, 1),
                                                          (1, 1, 1, 1, 0, 1), (1, 1, 0, 1, 0, 1), (0, 0, 1, 0, 0, 0), (0, 1, 1, 1, 1, 2)):
        ctx.set_flag("lanes_wanted", lanes_wanted)
        ctx.set_flag("side", side)
        ctx.set_flag("nm4", nm4)
        ctx.set_flag("early", early)
        ctx.set_flag("struct", struct)
        ctx.set_flag("exclusive", excl & 1)
        ctx.set_flag("background", excl >> 1)        # (excl = 2: the side lanes' ladders at wave priority 0, placement left alone)
        try:
            al, es, fs = sk.ProveDDLEQBatch(16, col("ct1"), col("ct2"), col("a"), col("b"), xs, ys)
        finally:
            ctx.set_flag("lanes_wanted", 0)
            ctx.set_flag("side", 1)
            ctx.set_flag("nm4", 1)
            ctx.set_flag("early", 1)
            ctx.set_flag("struct", 1)
            ctx.set_flag("exclusive", 1)
            ctx.set_flag("background", 0)
        for j in range(4):
            assert _digests(al[j], es[j], fs[j]) == [i["digest"] for i in by[j]], (lanes_wanted, side, nm4, early, struct, excl, j)


def test_ddleq_2048_kernels_off(ctx, keys):
    """The same fixture with the assembly / pair kernels switched off (hipcc VM kernels, ladders modulo n^3 instead of
    p^3, q^3): every implementation of the path must land on the committed numbers."""
    pk, sk = keys
    d = load("proofs.json")["ddleq"]
    st = [{k: int(v, 16) for k, v in s.items()} for s in d["statements"]]
    ins = d["instances"][:16]
    col = lambda key: [st[i["s"]][key] for i in ins]
    xs, ys = H(i["x"] for i in ins), H(i["y"] for i in ins)
    ctx.set_flag("pair", 0)
    try:
        al, es, fs = sk.ProveDDLEQInstancesBatch(col("ct1"), col("ct2"), col("a"), col("b"), xs, ys)
    finally:
        ctx.set_flag("pair", 1)
    assert (al, es, fs) == (H(i["alpha"] for i in ins), H(i["e"] for i in ins), H(i["f"] for i in ins))
    ctx.set_flag("asm", 0)
    try:
        assert pk.VerifyDDLEQInstancesBatch(col("ct1"), col("ct2"), xs, ys, al, es, fs) == [True] * len(ins)
    finally:
        ctx.set_flag("asm", 1)


def test_ddleq_2048_prover_with_the_p_adic_split_forced(ctx, keys):
    """A batch that fills the chip makes the prover split its ladders modulo p^3, q^3 by the base-p digits of the (reduced)
    exponents: x^(r0 + r1 p) = (x^(r1) mod p^2)^p x^(r0) (ddleq.cpp pow_n3_crt / pow_p2_multi_crt).  lanes_wanted = 1 forces that
    path -- one lane per number modulo p^2 -- for the 64 fixture instances; lanes_wanted = 4096 its two-lane variant; with
    the lift off the unsplit ladders run.  All must land on the committed Alpha / E / F."""
    pk, sk = keys
    d = load("proofs.json")["ddleq"]
    st = [{k: int(v, 16) for k, v in s.items()} for s in d["statements"]]
    ins = d["instances"]
    col = lambda key: [st[i["s"]][key] for i in ins]
    xs, ys = H(i["x"] for i in ins), H(i["y"] for i in ins)
    want = [i["digest"] for i in ins]
    for lanes_wanted, lift in ((1, 1), (4096, 1), (1, 0)):
        ctx.set_flag("lanes_wanted", lanes_wanted)
        ctx.set_flag("lift", lift)
        ctx.set_flag("struct", 0)         # (the ladders on ct1 itself: the default takes the structure path for Alpha)
        try:
            al, es, fs = sk.ProveDDLEQInstancesBatch(col("ct1"), col("ct2"), col("a"), col("b"), xs, ys)
        finally:
            ctx.set_flag("lanes_wanted", 0)
            ctx.set_flag("lift", 1)
            ctx.set_flag("struct", 1)
        dg = [hashlib.sha256(a.to_bytes(768, "big") + e.to_bytes(512, "big") + f.to_bytes(768, "big")).hexdigest()
              for a, e, f in zip(al, es, fs)]
        assert dg == want, (lanes_wanted, lift)


def test_encrypt_2048_on_the_full_batch_kernel(ctx, keys):
    """EncryptWithR at 2048 bits with lanes_wanted = 1: the batch keeps the kernel shape a 65 536-ciphertext batch uses
    (vm_asm_74_32, the two-lane pair kernel) instead of the small-batch re-slicing, and is compared with the committed c."""
    pk, sk = keys
    e = load("vectors.json")["2048"]["encrypt"]
    ctx.set_flag("lanes_wanted", 1)
    try:
        assert pk.EncryptWithRBatch(H(e["m"]), H(e["r"])) == H(e["c"])
        assert ctx.last_vm_asm() > 0
        d = load("vectors.json")["2048"]["decrypt"]
        assert sk.DecryptBatch(H(d["c"])) == H(d["m"])
        l2 = load("proofs.json")["level2"]
        import paillier_amd as pa
        assert pk.EncryptWithRBatch(H(l2["m"]), H(l2["r"]), level=pa.ENC_LEVEL_TWO) == H(l2["c"])
    finally:
        ctx.set_flag("lanes_wanted", 0)


def test_sub_and_level_two(ctx, keys):
    import paillier_amd as pa
    pk, sk = keys
    P = load("proofs.json")
    s = P["sub"]
    assert pk.SubBatch(H(s["l1"]["a"]), H(s["l1"]["b"])) == H(s["l1"]["out"])
    assert pk.SubBatch(H(s["l2"]["a"]), H(s["l2"]["b"]), level=pa.ENC_LEVEL_TWO) == H(s["l2"]["out"])
    a3, s3 = s["add3"], s["sub3"]
    assert pk.AddBatch(H(a3["a"]), H(a3["b"]), H(a3["c"])) == H(a3["out"])          # operations.go:11 variadic
    assert pk.SubBatch(H(s3["a"]), H(s3["b"]), H(s3["c"])) == H(s3["out"])          # operations.go:32 variadic
    one = s["sub1_unreduced"]
    assert pk.SubBatch([int(one["a"], 16)]) == [int(one["out"], 16)]                # single operand: returned as is
    l2 = P["level2"]
    cts = pk.EncryptWithRBatch(H(l2["m"]), H(l2["r"]), level=pa.ENC_LEVEL_TWO)
    assert cts == H(l2["c"])
    assert sk.DecryptBatch(cts, level=pa.ENC_LEVEL_TWO) == H(l2["m"])
    assert sk.DecryptBatch(H(l2["weird_c"]), level=pa.ENC_LEVEL_TWO) == H(l2["weird_m"])
    assert sk.DecryptBatch(H(l2["weird_c"]), level=pa.ENC_LEVEL_TWO, flags=pa.DECRYPT_NO_CRT) == H(l2["weird_m"])


def test_random_oracle_digest_fixture(ctx):
    """RandomOracleDigest (random_oracle.go:20-32) on the device: the caller drops argument 0, zero arguments are empty."""
    rows = load("proofs.json")["random_oracle"]
    for row in rows:
        args = H(row["args"])[1:]
        if not args:
            continue
        got = ctx.random_oracle_digest_batch([[a] * 3 for a in args])
        assert [g.hex() for g in got] == [row["digest"]] * 3
        assert got[0][-1] & 1 == row["bit"]


def test_threshold_fixture(ctx):
    import paillier_amd as pa
    P = load("proofs.json")
    t = load("keys.json")["threshold"]["2048"]
    tn, total, thr = int(t["n"], 16), t["total"], t["threshold"]
    shares, v, vks = H(t["shares"]), int(t["v"], 16), H(t["vks"])
    tk = pa.ThresholdPublicKey(ctx, tn, total=total, threshold=thr)
    th = P["threshold"]
    cs, parts = H(th["c"]), [H(r) for r in th["partials"]]
    for i in range(total):
        assert tk.PartialDecryptBatch(i + 1, shares[i], cs) == (i + 1, parts[i])
    for c in th["combine"]:
        assert tk.CombinePartialDecryptionsBatch([(i, parts[i - 1]) for i in c["ids"]]) == H(c["m"]), c["ids"]
    tam = th["tampered"]
    bad = [(i, [x ^ tam["xor_server3"] for x in parts[i - 1]] if i == 3 else parts[i - 1]) for i in tam["ids"]]
    assert tk.CombinePartialDecryptionsBatch(bad) == H(tam["m"])
    z = P["share_zkp"]
    sid, recs = z["server"], z["proofs"]
    cts, rs = H(r["c"] for r in recs), H(r["r"] for r in recs)
    dec, es, zs = tk.PartialDecryptionWithZKPBatch(sid, shares[sid - 1], v, cts, rs)
    assert dec == H(r["dec"] for r in recs)
    assert es == H(r["e"] for r in recs)
    assert zs == H(r["z"] for r in recs)
    assert tk.VerifyProofBatch(v, vks[sid - 1], cts, dec, es, zs) == [True] * len(recs)
    # the two hashed residues of the verifier (thresholdkey.go:294-311), recomputed from GPU primitives
    m2 = pa.Modulus(ctx, tn * tn)
    c4 = tk.ConstMultBatch(cts, 4)
    a = m2.mul_batch(m2.exp_batch(c4, zs), m2.inv_batch(m2.exp_batch(m2.mul_batch(dec, dec), es)))
    assert a == H(r["verify_a"] for r in recs)
    digests = ctx.random_oracle_digest_batch([H(r["a"] for r in recs), H(r["b"] for r in recs), H(r["c4"] for r in recs),
                                              H(r["ci2"] for r in recs)])
    assert [int.from_bytes(g, "big") for g in digests] == es
