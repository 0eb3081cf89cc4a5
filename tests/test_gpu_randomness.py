"""The batched randomness pipeline: PublicKey.Encrypt / EncryptAtLevel draw r in Z_n^* themselves
(paillier.go:258-269, utils.go:26-49).  Bit-exact comparison is only possible through the returned r: the ciphertext must be
EncryptWithR(m, r) for exactly that r, and r must be a valid draw (0 < r < n, gcd(r, n) = 1, no repeats)."""
import math
import random

import numpy as np
import pytest

from oracle import paillier_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import paillier_amd as pa
    return pa.Context(0)


@pytest.mark.parametrize("bits,level", [(1024, 0), (2048, 0), (1024, 1)])
def test_encrypt_draws_its_own_randomness(ctx, bits, level):
    import paillier_amd as pa
    sk_o, p, q = po.keygen_seeded(bits, bits + 7)
    n = sk_o.N
    pk = pa.PublicKey(ctx, n, n + 1)
    sk = pa.SecretKey(ctx, pk, sk_o.Lambda)
    rng = random.Random(1)
    B = 300
    top = n if level == 0 else n * n
    ms = [0, 1, top - 1] + [rng.randrange(top) for _ in range(B - 3)]
    cts, rs = pk.EncryptBatch(ms, level=level, return_r=True)
    assert all(0 < r < n and math.gcd(r, n) == 1 for r in rs)
    assert len(set(rs)) == B and len(set(cts)) == B
    assert cts == pk.EncryptWithRBatch(ms, rs, level=level)                       # the ciphertext of exactly that r
    assert cts[:4] == [po.encrypt_with_r_at_level(sk_o, m, r, level).C for m, r in zip(ms[:4], rs[:4])]
    assert sk.DecryptBatch(cts, level=level) == ms                                 # paillier_test.go:52-63
    again = pk.EncryptBatch(ms, level=level)
    assert not set(again) & set(cts)                                               # fresh randomness every call


def test_random_units_rejects_non_units(ctx):
    """A toy modulus with many non-units (n = 3 * 5 * 7 is not a Paillier modulus, but the sampler only needs an odd n): every
    value returned must be a unit, which exercises the redraw path (device GCD flags -> host redraw) on nearly every call; and
    the reference's 101 * 103 (thresholdkey_test.go:58) where ~2 % of the draws are rejected."""
    import paillier_amd as pa
    for n in (105, 101 * 103):
        pk = pa.PublicKey(ctx, n, n + 1)
        rs = pk.random_units(5000)
        assert all(0 < r < n and math.gcd(r, n) == 1 for r in rs)
        units = [r for r in range(1, n) if math.gcd(r, n) == 1]
        counts = np.bincount(rs, minlength=n)
        seen = counts[units]
        if n == 105:                                   # 48 units, ~104 draws each: a crude uniformity check (6 sigma)
            assert seen.min() > 104 - 62 and seen.max() < 104 + 62
        assert counts.sum() == 5000 and counts[[r for r in range(n) if r not in set(units)]].sum() == 0


@pytest.mark.parametrize("side", [1, 0])
@pytest.mark.parametrize("n", [105, 101 * 103])
def test_encrypt_redoes_the_call_when_a_draw_is_not_a_unit(n, side):
    """pgpu_encrypt only STARTS the gcd test of its draws before the ladder (side stream) and asks for the verdict afterwards; if a
    draw was not a unit the whole call is redone with the test first (careful redraw).  For an honest modulus that branch never
    runs; toy moduli with many non-units (105: 54 % of the draws; 101 * 103, thresholdkey_test.go:58: 2 %) take it on every call.
    What comes back must be the SECOND attempt's outputs: every r a unit (utils.go:43) and c the ciphertext of exactly that r.
    side = 0: no side stream, the test runs first."""
    import paillier_amd as pa
    ctx = pa.Context(0)
    ctx.set_flag("side", side)
    pk = pa.PublicKey(ctx, n, n + 1)
    rng = random.Random(n + side)
    B = 700
    ms = [0, 1, n - 1] + [rng.randrange(n) for _ in range(B - 3)]
    for _ in range(3):
        cts, rs = pk.EncryptBatch(ms, return_r=True)
        assert len(cts) == len(rs) == B
        assert all(0 < r < n and math.gcd(r, n) == 1 for r in rs), "a non-unit r came back: the first attempt's buffer was returned"
        n2 = n * n
        assert cts == [(1 + m * n) % n2 * pow(r, n, n2) % n2 for m, r in zip(ms, rs)]       # paillier.go:206-218 with G = n + 1
        assert cts == pk.EncryptWithRBatch(ms, rs)
    assert len({tuple(pk.EncryptBatch(ms[:50], return_r=True)[1]) for _ in range(3)}) == 3   # fresh draws every call
