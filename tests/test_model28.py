"""CPU test of the radix-2^28 multi-lane Montgomery model (tests/model28.py), the algorithm both GPU kernels transcribe:
value correctness and the 64-bit accumulator bounds (every accumulator write is asserted < 2^64) for every built shape,
including the all-ones worst case and lazy (2^28 + 1) limbs."""
import random

import pytest

from model28 import MASK, from_limbs, mont_consts, montmul_model, to_limbs

SHAPES = [(1024, 37, 1), (1536, 55, 1), (2048, 74, 1), (2048, 37, 2), (3072, 55, 2), (4096, 74, 2), (4096, 37, 4), (6144, 55, 4)]


@pytest.mark.parametrize("bits,wl,k", SHAPES)
def test_montmul_model(bits, wl, k):
    rng = random.Random(bits * 10 + k)
    wt = wl * k
    for n in (rng.getrandbits(bits) | (1 << (bits - 1)) | 1, (1 << bits) - 159):
        mc = mont_consts(n, wt)
        for a, b in [(2 * n - 1, 2 * n - 1), (rng.randrange(2 * n), rng.randrange(2 * n)), (0, 5), (1, n - 1)]:
            out = montmul_model(to_limbs(b, wt), to_limbs(a, wt), mc["n"], mc["n0inv"], wl, k)
            v = from_limbs(out)
            assert v < 2 * n and (v * mc["R"] - a * b) % n == 0
    # lazy operand limbs (a limb of 0 borrows 2^28 from its neighbour): same value, limbs up to 2^28 + 1
    n = (1 << bits) - 159
    mc = mont_consts(n, wt)
    a = 2 * n - 1
    lz = to_limbs(a, wt)
    for j in range(wt - 1):
        if lz[j] <= 1 and lz[j + 1] >= 1:
            lz[j] += MASK + 1
            lz[j + 1] -= 1
    assert from_limbs(lz) == a
    v = from_limbs(montmul_model(lz, lz, mc["n"], mc["n0inv"], wl, k))
    assert v < 2 * n and (v * mc["R"] - a * a) % n == 0
