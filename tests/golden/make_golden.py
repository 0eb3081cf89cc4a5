"""Generates the committed golden fixtures (tests/golden/*.json) from the CPU oracle.

Run in the build container:  python tests/golden/make_golden.py
The reference is Go and cannot be imported or run here; the vectors therefore come from the
Python-int oracle (oracle/paillier_oracle.py), which is pinned to the reference's own KATs
(tests/test_oracle_kats.py) and cross-checked against libgmp (tests/test_oracle_cross.py).
Every value is a hex string.  Seeds are fixed, so re-running reproduces the files byte for byte.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import paillier_oracle as po  # noqa: E402


def hx(v):
    return format(int(v), "x")


def extra_4096():
    """A 4096-bit key (2048-bit primes: 74-limb CRT halves, the two-lane pair kernel's widest use) with a few Decrypt
    vectors, in a file of its own so that the older fixtures stay byte-identical."""
    sk, p, q = po.keygen_seeded(4096, 4096)
    rng = random.Random(4196)
    n, n2 = sk.N, sk.N ** 2
    ms = [0, 1, n - 1] + [rng.randrange(n) for _ in range(5)]
    rs = [po.rand_unit(n, rng) for _ in ms]
    cts = [po.encrypt_with_r(sk, m, r).C for m, r in zip(ms, rs)]
    weird = [0, p, 7 * q, n, n2 - 1] + [rng.randrange(n2) for _ in range(3)]
    out = {"p": hx(p), "q": hx(q), "n": hx(n), "lambda": hx(sk.Lambda), "m": [hx(m) for m in ms], "r": [hx(r) for r in rs],
           "c": [hx(c) for c in cts], "weird_c": [hx(c) for c in weird],
           "weird_m": [hx(po.decrypt(sk, po.Ciphertext(c))) for c in weird]}
    with open(os.path.join(HERE, "key4096.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def fixed(v, nbytes):
    return int(v).to_bytes(nbytes, "big")


def proofs_fixture():
    """SURVEY.md 8(c) "fixtures to commit" beyond Encrypt/Decrypt/Add/ConstMult, at the BASELINE key size (2048 bits):
    Sub, level-two Encrypt/Decrypt, RandomOracleDigest quirks, the share-decryption ZKP with fixed r (including the two
    hashed residues a, b), Combine for every 3-subset of 5 servers, DDLEQ prove (x, y supplied, both challenge bits) and
    verify (valid and wrong-ct2).  Keys come from keys.json; a file of its own so that the older fixtures stay
    byte-identical.  Takes about two minutes of Python-int arithmetic."""
    import hashlib
    import itertools
    K = json.load(open(os.path.join(HERE, "keys.json")))
    k = K["paillier"]["2048"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    sk = po.SecretKey(N=p * q, G=p * q + 1, Lambda=(p - 1) * (q - 1))
    n, n2, n3 = sk.N, sk.N ** 2, sk.N ** 3
    nb1, nb2, nb3 = (n.bit_length() + 7) // 8, (n2.bit_length() + 7) // 8, (n3.bit_length() + 7) // 8
    L1, L2 = po.ENC_LEVEL_ONE, po.ENC_LEVEL_TWO
    out = {"key_bits": 2048}

    # ---- Sub (operations.go:32-55) at both levels, two operands
    rng = random.Random(7001)
    a1 = [rng.randrange(n2) for _ in range(4)]
    b1 = [po.encrypt_with_r(sk, rng.randrange(n), po.rand_unit(n, rng)).C for _ in range(3)] + [n2 - 1]
    a2 = [rng.randrange(n3) for _ in range(3)]
    b2 = [po.encrypt_with_r_at_level(sk, rng.randrange(n2), po.rand_unit(n, rng), L2).C for _ in range(3)]
    out["sub"] = {
        "l1": {"a": [hx(v) for v in a1], "b": [hx(v) for v in b1],
               "out": [hx(po.sub(sk, po.Ciphertext(x), po.Ciphertext(y)).C) for x, y in zip(a1, b1)]},
        "l2": {"a": [hx(v) for v in a2], "b": [hx(v) for v in b2],
               "out": [hx(po.sub(sk, po.Ciphertext(x, L2), po.Ciphertext(y, L2)).C) for x, y in zip(a2, b2)]},
        # variadic forms (operations.go:11,32): Add() of three operands; Sub with ONE operand returns it unreduced
        "add3": {"a": [hx(v) for v in a1[:3]], "b": [hx(v) for v in b1[:3]], "c": [hx(v) for v in a1[1:4]],
                 "out": [hx(po.add(sk, po.Ciphertext(x), po.Ciphertext(y), po.Ciphertext(z)).C)
                         for x, y, z in zip(a1[:3], b1[:3], a1[1:4])]},
        "sub3": {"a": [hx(v) for v in a1[:3]], "b": [hx(v) for v in b1[:3]], "c": [hx(v) for v in b1[1:4]],
                 "out": [hx(po.sub(sk, po.Ciphertext(x), po.Ciphertext(y), po.Ciphertext(z)).C)
                         for x, y, z in zip(a1[:3], b1[:3], b1[1:4])]},
        "sub1_unreduced": {"a": hx(n2 + 5), "out": hx(po.sub(sk, po.Ciphertext(n2 + 5)).C)},
    }

    # ---- level-two Encrypt / Decrypt (paillier.go:206-218,292-340)
    rng = random.Random(7002)
    ms = [0, 1, n - 1, n, n2 - 1] + [rng.randrange(n2) for _ in range(3)]
    rs = [po.rand_unit(n, rng) for _ in ms]
    cts = [po.encrypt_with_r_at_level(sk, m, r, L2).C for m, r in zip(ms, rs)]
    weird = [0, p, n, n2, n3 - 1, rng.randrange(n3)]
    out["level2"] = {"m": [hx(v) for v in ms], "r": [hx(v) for v in rs], "c": [hx(v) for v in cts],
                     "weird_c": [hx(v) for v in weird],
                     "weird_m": [hx(po.decrypt(sk, po.Ciphertext(c, L2))) for c in weird]}

    # ---- RandomOracleDigest / RandomOracleBit (random_oracle.go:10-32): arg 0 skipped, zero -> no bytes
    rng = random.Random(7003)
    rows = [[5, 2, 3], [999, 2, 3], [1, 0, 3], [1, 3], [7, 0, 0, 0], [0], [rng.getrandbits(6144), rng.getrandbits(6144), 0,
            rng.getrandbits(2048), rng.getrandbits(6100)], [1, 1 << 2047, 255, 256, 0, 1]]
    out["random_oracle"] = [{"args": [hx(v) for v in row], "digest": po.random_oracle_digest(*row).hex(),
                             "bit": int(po.random_oracle_bit(*row))} for row in rows]

    # ---- threshold: PartialDecrypt for all 5 servers, Combine for every 3-subset, share ZKP with fixed r
    t = K["threshold"]["2048"]
    tn, total, thr = int(t["n"], 16), t["total"], t["threshold"]
    shares = [int(s, 16) for s in t["shares"]]
    v, vks = int(t["v"], 16), [int(x, 16) for x in t["vks"]]
    tsks = [po.ThresholdSecretKey(N=tn, G=tn + 1, TotalNumberOfDecryptionServers=total, Threshold=thr, VerificationKey=v,
                                  VerificationKeys=vks, ID=i + 1, Share=shares[i]) for i in range(total)]
    rng = random.Random(7004)
    tms = [0, tn - 1, rng.randrange(tn), rng.randrange(tn)]
    tcs = [po.encrypt_with_r(tsks[0], m, po.rand_unit(tn, rng)).C for m in tms]
    parts = [[po.partial_decrypt(ts, c).Decryption for c in tcs] for ts in tsks]
    comb = []
    for ids in list(itertools.combinations(range(1, total + 1), thr)) + [(5, 3, 1), (1, 2, 3, 4, 5)]:
        comb.append({"ids": list(ids),
                     "m": [hx(po.combine_partial_decryptions(tsks[0], [po.PartialDecryption(i, parts[i - 1][j]) for i in ids]))
                           for j in range(len(tcs))]})
    bad = [(1, parts[0]), (3, [x ^ 5 for x in parts[2]]), (4, parts[3])]
    out["threshold"] = {"m": [hx(v_) for v_ in tms], "c": [hx(v_) for v_ in tcs],
                        "partials": [[hx(x) for x in row] for row in parts], "combine": comb,
                        "tampered": {"ids": [1, 3, 4], "xor_server3": 5,
                                     "m": [hx(po.combine_partial_decryptions(tsks[0], [po.PartialDecryption(i, d[j]) for i, d in bad]))
                                           for j in range(len(tcs))]}}
    sid = 2
    tn2 = tn * tn
    zr = [0, 1, tn2 - 1] + [rng.randrange(tn2) for _ in range(5)]
    zc = [tcs[i % len(tcs)] for i in range(3)] + [po.encrypt_with_r(tsks[0], rng.randrange(tn), po.rand_unit(tn, rng)).C for _ in range(5)]
    zk = []
    for c, r in zip(zc, zr):
        pf = po.partial_decryption_with_zkp_r(tsks[sid - 1], c, r)
        zk.append({"c": hx(c), "r": hx(r), "dec": hx(pf.Decryption), "e": hx(pf.E), "z": hx(pf.Z),
                   # the four hashed integers (thresholdkey.go:241-252): a, b reduced; c^4, dec^2 UNREDUCED
                   "a": hx(po.gmp_exp(c ** 4, r, tn2)), "b": hx(po.gmp_exp(v, r, tn2)), "c4": hx(c ** 4), "ci2": hx(pf.Decryption ** 2),
                   "verify_a": hx(po.verify_part1(pf)), "verify_b": hx(po.verify_part2(pf))})
    out["share_zkp"] = {"server": sid, "proofs": zk}

    # ---- DDLEQ (ddleq.go:55-153): 4 statements x 16 instances, draws supplied
    rng = random.Random(7005)
    stmts = []
    for _ in range(4):
        m = rng.randrange(n)
        inner = po.encrypt_with_r(sk, m, po.rand_unit(n, rng)).C
        ct1 = po.encrypt_with_r_at_level(sk, inner, po.rand_unit(n, rng), L2).C
        a, b = po.rand_unit(n, rng), po.rand_unit(n, rng)
        ct2 = po.nested_randomize_with_ab(sk, po.Ciphertext(ct1, L2), a, b).C
        stmts.append({"ct1": ct1, "ct2": ct2, "a": a, "b": b})
    inst = []
    for i in range(64):
        s = stmts[i % 4]
        x, y = po.rand_unit(n, rng), po.rand_unit(n, rng)
        pf = po.prove_ddleq_instance_xy(sk, po.Ciphertext(s["ct1"], L2), po.Ciphertext(s["ct2"], L2), s["a"], s["b"], x, y)
        bit = int(po.random_oracle_bit(s["ct1"], s["ct2"], x, y, pf.Alpha))
        assert po.verify_ddleq_proof_instance(sk, po.Ciphertext(s["ct1"], L2), po.Ciphertext(s["ct2"], L2), pf)
        wrong = stmts[(i + 1) % 4]["ct2"]
        rec = {"s": i % 4, "x": hx(x), "y": hx(y), "bit": bit,
               "digest": hashlib.sha256(fixed(pf.Alpha, nb3) + fixed(pf.E, nb2) + fixed(pf.F, nb3)).hexdigest(),
               "verify_wrong_ct2": int(po.verify_ddleq_proof_instance(sk, po.Ciphertext(s["ct1"], L2), po.Ciphertext(wrong, L2), pf))}
        if i < 16:
            rec.update({"alpha": hx(pf.Alpha), "e": hx(pf.E), "f": hx(pf.F)})
        inst.append(rec)
        print("ddleq instance", i, "bit", bit, flush=True)
    assert 0 < sum(r["bit"] for r in inst) < 64
    out["ddleq"] = {"statements": [{kk: hx(vv) for kk, vv in s.items()} for s in stmts], "instances": inst,
                    "digest_of": "sha256(alpha[768 B] || e[512 B] || f[768 B]), big-endian fixed width",
                    "wrong_ct2": "ct2 of statement (s + 1) mod 4"}
    with open(os.path.join(HERE, "proofs.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote proofs.json")


def secpar_fixture():
    """ProveDDLEQ (ddleq.go:27-40) at the reference's test setting secpar = 40 (ddleq_test.go:74-88): ONE statement -- statement 0
    of proofs.json -- with 40 instances, draws supplied, and statement 1 with 40 more (the second statement pins the
    statement-major row order of a batch of statements).
    Each instance is the oracle's proveDDLEQInstance on its own, i.e. what the reference computes per instance WITHOUT hoisting.
    A file of its own so that the older fixtures stay byte-identical."""
    import hashlib
    K = json.load(open(os.path.join(HERE, "keys.json")))
    k = K["paillier"]["2048"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    sk = po.SecretKey(N=p * q, G=p * q + 1, Lambda=(p - 1) * (q - 1))
    n, n2, n3 = sk.N, sk.N ** 2, sk.N ** 3
    nb2, nb3 = (n2.bit_length() + 7) // 8, (n3.bit_length() + 7) // 8
    L2 = po.ENC_LEVEL_TWO
    st = json.load(open(os.path.join(HERE, "proofs.json")))["ddleq"]["statements"]
    rng = random.Random(7040)
    out = {"key_bits": 2048, "secpar": 40, "statements_of": "proofs.json ddleq.statements[0], [1]",
           "digest_of": "sha256(alpha[768 B] || e[512 B] || f[768 B]), big-endian fixed width", "proofs": []}
    for j in (0, 1):
        s = {kk: int(vv, 16) for kk, vv in st[j].items()}
        inst = []
        for i in range(40):
            x, y = po.rand_unit(n, rng), po.rand_unit(n, rng)
            pf = po.prove_ddleq_instance_xy(sk, po.Ciphertext(s["ct1"], L2), po.Ciphertext(s["ct2"], L2), s["a"], s["b"], x, y)
            assert po.verify_ddleq_proof_instance(sk, po.Ciphertext(s["ct1"], L2), po.Ciphertext(s["ct2"], L2), pf)
            rec = {"x": hx(x), "y": hx(y), "bit": int(po.random_oracle_bit(s["ct1"], s["ct2"], x, y, pf.Alpha)),
                   "digest": hashlib.sha256(fixed(pf.Alpha, nb3) + fixed(pf.E, nb2) + fixed(pf.F, nb3)).hexdigest()}
            if i < 4:
                rec.update({"alpha": hx(pf.Alpha), "e": hx(pf.E), "f": hx(pf.F)})
            inst.append(rec)
            print("secpar fixture: statement", j, "instance", i, "bit", rec["bit"], flush=True)
        out["proofs"].append({"statement": j, "instances": inst})
    with open(os.path.join(HERE, "ddleq_secpar40.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote ddleq_secpar40.json")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "4096":
        return extra_4096()
    if len(sys.argv) > 1 and sys.argv[1] == "secpar":
        return secpar_fixture()
    if len(sys.argv) > 1 and sys.argv[1] == "proofs":
        return proofs_fixture()
    keys = {}
    for bits in (1024, 2048, 3072):
        sk, p, q = po.keygen_seeded(bits, bits)
        keys[str(bits)] = {"p": hx(p), "q": hx(q), "n": hx(sk.N), "g": hx(sk.G), "h": hx(sk.H), "k": hx(sk.K),
                           "lambda": hx(sk.Lambda)}
    # threshold keys with genuine safe primes (thresholdkey_generator.go:47-56); l=5, t=3 as in BASELINE config 4
    tkeys = {}
    for bits in (512, 2048):
        rng = random.Random(4000 + bits)
        while True:
            p, p1 = po.gen_safe_prime(bits // 2, rng)
            q, q1 = po.gen_safe_prime(bits // 2, rng)
            if p != q and p != q1 and p1 != q and (p * q).bit_length() == bits:
                break
        tsks = po.threshold_keys_from_primes(p, p1, q, q1, 5, 3, rng)
        tkeys[str(bits)] = {"p": hx(p), "p1": hx(p1), "q": hx(q), "q1": hx(q1), "n": hx(tsks[0].N),
                            "total": 5, "threshold": 3, "v": hx(tsks[0].VerificationKey),
                            "vks": [hx(v) for v in tsks[0].VerificationKeys], "shares": [hx(t.Share) for t in tsks]}
    with open(os.path.join(HERE, "keys.json"), "w") as f:
        json.dump({"paillier": keys, "threshold": tkeys}, f, indent=0, sort_keys=True)

    vec = {}
    for bits in (1024, 2048, 3072):
        k = keys[str(bits)]
        sk = po.key_from_primes(int(k["p"], 16), int(k["q"], 16), 1)
        sk.H = int(k["h"], 16)
        rng = random.Random(100 + bits)
        n, n2 = sk.N, sk.N ** 2
        ms = [0, 1, n - 1] + [rng.randrange(n) for _ in range(5)]
        rs = [po.rand_unit(n, rng) for _ in ms]
        cts = [po.encrypt_with_r(sk, m, r).C for m, r in zip(ms, rs)]
        # arbitrary elements of Z_{n^2}, including non-units (gcd(c, n) != 1) and zero
        weird = [0, int(k["p"], 16), 7 * int(k["q"], 16), n, n2 - 1] + [rng.randrange(n2) for _ in range(3)]
        a = [rng.randrange(n2) for _ in range(4)]
        b = [rng.randrange(n2) for _ in range(4)]
        kk = [50 ** 50 % n] + [rng.randrange(n) for _ in range(3)]
        vec[str(bits)] = {
            "encrypt": {"m": [hx(v) for v in ms], "r": [hx(v) for v in rs], "c": [hx(v) for v in cts]},
            "decrypt": {"c": [hx(v) for v in cts + weird],
                        "m": [hx(po.decrypt(sk, po.Ciphertext(c))) for c in cts + weird]},
            "add": {"a": [hx(v) for v in a], "b": [hx(v) for v in b],
                    "out": [hx(po.add(sk, po.Ciphertext(x), po.Ciphertext(y)).C) for x, y in zip(a, b)]},
            "const_mult": {"c": [hx(v) for v in a], "k": [hx(v) for v in kk],
                           "out": [hx(po.const_mult(sk, po.Ciphertext(x), e).C) for x, e in zip(a, kk)],
                           "out_shared_k0": [hx(po.const_mult(sk, po.Ciphertext(x), kk[0]).C) for x in a]},
        }
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(vec, f, indent=0, sort_keys=True)
    print("wrote keys.json, vectors.json")


if __name__ == "__main__":
    main()
