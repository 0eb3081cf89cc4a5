"""tests/golden/proofs.json (Sub, level two, RandomOracleDigest, threshold Combine, share ZKP, DDLEQ at 2048 bits; written by
make_golden.py from the Python-int oracle) must reproduce from the oracles on the CPU: the libgmp restatement proves and
verifies ALL 64 DDLEQ instances again (a second, independent implementation of ddleq.go:55-153 incl. SHA-256), the
Python-int oracle re-derives a sample of every other section.  Guards against fixture rot."""
import hashlib
import itertools
import json
import os

from oracle import gmp_oracle as go
from oracle import paillier_oracle as po

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
L2 = po.ENC_LEVEL_TWO


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def H(xs):
    return [int(x, 16) for x in xs]


def key2048():
    k = load("keys.json")["paillier"]["2048"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    return po.SecretKey(N=p * q, G=p * q + 1, Lambda=(p - 1) * (q - 1)), p, q


def test_ddleq_fixture_reproduces_from_libgmp():
    sk, p, q = key2048()
    n = sk.N
    d = load("proofs.json")["ddleq"]
    st = [{k: int(v, 16) for k, v in s.items()} for s in d["statements"]]
    ins = d["instances"]
    assert len(ins) == 64 and 0 < sum(i["bit"] for i in ins) < 64
    col = lambda key: [st[i["s"]][key] for i in ins]
    xs, ys = H(i["x"] for i in ins), H(i["y"] for i in ins)
    al, es, fs, bits = go.ddleq_prove_batch(n, sk.Lambda, col("ct1"), col("ct2"), col("a"), col("b"), xs, ys, threads=8)
    assert bits == [i["bit"] for i in ins]
    dg = [hashlib.sha256(a.to_bytes(768, "big") + e.to_bytes(512, "big") + f.to_bytes(768, "big")).hexdigest()
          for a, e, f in zip(al, es, fs)]
    assert dg == [i["digest"] for i in ins]
    for i, rec in enumerate(ins[:16]):
        assert (al[i], es[i], fs[i]) == (int(rec["alpha"], 16), int(rec["e"], 16), int(rec["f"], 16))
        if not rec["bit"]:
            assert (es[i], fs[i]) == (xs[i], ys[i])          # ddleq.go:94,101: e = x, f = y when the challenge bit is 0
    assert go.ddleq_verify_batch(n, col("ct1"), col("ct2"), xs, ys, al, es, fs, threads=8) == [True] * 64
    wrong = [st[(i["s"] + 1) % 4]["ct2"] for i in ins]
    assert go.ddleq_verify_batch(n, col("ct1"), wrong, xs, ys, al, es, fs, threads=8) == [bool(i["verify_wrong_ct2"]) for i in ins]
    # one instance of each challenge bit through the Python-int oracle as well
    for want_bit in (0, 1):
        i = next(j for j, r in enumerate(ins[:16]) if r["bit"] == want_bit)
        s = st[ins[i]["s"]]
        pf = po.prove_ddleq_instance_xy(sk, po.Ciphertext(s["ct1"], L2), po.Ciphertext(s["ct2"], L2), s["a"], s["b"], xs[i], ys[i])
        assert (pf.Alpha, pf.E, pf.F) == (al[i], es[i], fs[i])


def test_ddleq_secpar40_fixture_reproduces_from_libgmp():
    """tests/golden/ddleq_secpar40.json (make_golden.py secpar): ProveDDLEQ at the reference's test setting secpar = 40
    (ddleq_test.go:74-88) for two statements -- each instance is proveDDLEQInstance on its own in both oracles (no hoisting)."""
    sk, p, q = key2048()
    n = sk.N
    d = load("ddleq_secpar40.json")
    st = [{k: int(v, 16) for k, v in s.items()} for s in load("proofs.json")["ddleq"]["statements"]]
    assert d["secpar"] == 40 and [pr["statement"] for pr in d["proofs"]] == [0, 1]
    for pr in d["proofs"]:
        s, ins = st[pr["statement"]], pr["instances"]
        assert len(ins) == 40 and 0 < sum(i["bit"] for i in ins) < 40
        xs, ys = H(i["x"] for i in ins), H(i["y"] for i in ins)
        rep = lambda key: [s[key]] * 40
        al, es, fs, bits = go.ddleq_prove_batch(n, sk.Lambda, rep("ct1"), rep("ct2"), rep("a"), rep("b"), xs, ys, threads=8)
        assert bits == [i["bit"] for i in ins]
        dg = [hashlib.sha256(a.to_bytes(768, "big") + e.to_bytes(512, "big") + f.to_bytes(768, "big")).hexdigest()
              for a, e, f in zip(al, es, fs)]
        assert dg == [i["digest"] for i in ins]
        for i, rec in enumerate(ins[:4]):
            assert (al[i], es[i], fs[i]) == (int(rec["alpha"], 16), int(rec["e"], 16), int(rec["f"], 16))
        assert all(go.ddleq_verify_batch(n, rep("ct1"), rep("ct2"), xs, ys, al, es, fs, threads=8))


def test_sub_level2_random_oracle_reproduce():
    sk, p, q = key2048()
    P = load("proofs.json")
    s = P["sub"]
    assert [po.sub(sk, po.Ciphertext(a), po.Ciphertext(b)).C for a, b in zip(H(s["l1"]["a"]), H(s["l1"]["b"]))] == H(s["l1"]["out"])
    assert [po.sub(sk, po.Ciphertext(a, L2), po.Ciphertext(b, L2)).C
            for a, b in zip(H(s["l2"]["a"]), H(s["l2"]["b"]))] == H(s["l2"]["out"])
    a3 = s["add3"]
    assert [po.add(sk, *(po.Ciphertext(v) for v in t)).C for t in zip(H(a3["a"]), H(a3["b"]), H(a3["c"]))] == H(a3["out"])
    s3 = s["sub3"]
    assert [po.sub(sk, *(po.Ciphertext(v) for v in t)).C for t in zip(H(s3["a"]), H(s3["b"]), H(s3["c"]))] == H(s3["out"])
    assert int(s["sub1_unreduced"]["out"], 16) == int(s["sub1_unreduced"]["a"], 16) > sk.N ** 2   # operations.go:37: not reduced
    l2 = P["level2"]
    got = [po.encrypt_with_r_at_level(sk, m, r, L2).C for m, r in zip(H(l2["m"])[:3], H(l2["r"])[:3])]
    assert got == H(l2["c"])[:3]
    # libgmp: r^(n^2) mod n^3 for every vector, times the closed form of (1+n)^m
    n, n2, n3 = sk.N, sk.N ** 2, sk.N ** 3
    rn = go.modexp_batch(n3, n2, H(l2["r"]), threads=4)
    assert [pow(n + 1, m, n3) * x % n3 for m, x in zip(H(l2["m"]), rn)] == H(l2["c"])
    assert [po.decrypt(sk, po.Ciphertext(c, L2)) for c in H(l2["weird_c"])[:3]] == H(l2["weird_m"])[:3]
    for row in P["random_oracle"]:
        args = H(row["args"])
        assert po.random_oracle_digest(*args).hex() == row["digest"] and int(po.random_oracle_bit(*args)) == row["bit"]
    ro = P["random_oracle"]
    assert ro[0]["digest"] == ro[1]["digest"]           # argument 0 does not enter the hash
    assert ro[2]["digest"] == ro[3]["digest"]           # a zero argument contributes no bytes
    assert ro[4]["digest"] == ro[5]["digest"] == hashlib.sha256(b"").hexdigest()


def test_threshold_and_share_zkp_reproduce():
    P = load("proofs.json")
    t = load("keys.json")["threshold"]["2048"]
    tn, total, thr = int(t["n"], 16), t["total"], t["threshold"]
    shares, v, vks = H(t["shares"]), int(t["v"], 16), H(t["vks"])
    tsks = [po.ThresholdSecretKey(N=tn, G=tn + 1, TotalNumberOfDecryptionServers=total, Threshold=thr, VerificationKey=v,
                                  VerificationKeys=vks, ID=i + 1, Share=shares[i]) for i in range(total)]
    th = P["threshold"]
    cs, parts = H(th["c"]), [H(r) for r in th["partials"]]
    # PartialDecrypt on libgmp (thresholdkey.go:192-201): c^(2 l! s_i) mod n^2
    for ts, row in zip(tsks, parts):
        assert go.modexp_batch(tn * tn, ts.Share * 2 * po.factorial(total), cs, threads=4) == row
    subsets = [c["ids"] for c in th["combine"]]
    assert [tuple(s) for s in subsets[:10]] == list(itertools.combinations(range(1, 6), 3))
    for c in th["combine"]:
        got = [po.combine_partial_decryptions(tsks[0], [po.PartialDecryption(i, parts[i - 1][j]) for i in c["ids"]])
               for j in range(len(cs))]
        assert got == H(c["m"]) == H(th["m"])
    z = P["share_zkp"]
    sid = z["server"]
    for rec in z["proofs"][:4]:
        c, r = int(rec["c"], 16), int(rec["r"], 16)
        pf = po.partial_decryption_with_zkp_r(tsks[sid - 1], c, r)
        assert (pf.Decryption, pf.E, pf.Z) == (int(rec["dec"], 16), int(rec["e"], 16), int(rec["z"], 16))
        assert po.compute_hash(int(rec["a"], 16), int(rec["b"], 16), int(rec["c4"], 16), int(rec["ci2"], 16)) == pf.E
        assert int(rec["c4"], 16) == c ** 4 and int(rec["ci2"], 16) == pf.Decryption ** 2
        assert po.verify_proof(pf) and po.verify_part1(pf) == int(rec["verify_a"], 16) == int(rec["a"], 16)
        assert po.verify_part2(pf) == int(rec["verify_b"], 16) == int(rec["b"], 16)
