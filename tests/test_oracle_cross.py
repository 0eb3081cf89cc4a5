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
