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
