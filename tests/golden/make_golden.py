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


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "4096":
        return extra_4096()
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
