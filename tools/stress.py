#!/usr/bin/env python3
"""Large-sample round-trip / identity checks on the GPU for paths whose unit tests use few samples (rare data-dependent
bugs -- e.g. a lazy value landing in [N, 2N) -- only show at ~1e-4 rates).  Properties are size-independent:
Decrypt(Encrypt(m)) == m at both levels, (a*b)*b^-1 == a, x*x^-1 == 1, homomorphic identities, threshold round trip."""
import json, os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import paillier_amd as pa

K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))
ctx = pa.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(99)
for bits, natural in ((1024, 0), (2048, 0), (2048, 1), (3072, 1)):
    # natural = 1: every modulus on its natural kernel shape (n^2 of the 2048-bit key: the wave-sliced 148-limb kernel)
    ctx.set_flag("lanes_wanted", natural)
    k = K["paillier"][str(bits)]
    p, q = int(k["p"], 16), int(k["q"], 16)
    n, lam = p * q, (p - 1) * (q - 1)
    n2 = n * n
    pk = pa.PublicKey(ctx, n, n + 1, H=int(k["h"], 16), K=int(k["k"], 16))
    sk = pa.SecretKey(ctx, pk, lam)
    t = time.time()
    ms = [rng.randrange(n) for _ in range(B)]
    rs = [rng.randrange(1, n) for _ in range(B)]
    c1 = pk.EncryptWithRBatch(ms, rs)
    assert sk.DecryptBatch(c1) == ms, "L1 round trip"
    assert sk.DecryptBatch(c1, flags=1) == ms, "L1 round trip, reference formula"
    ca, _ = pk.AltEncryptWithRBatch(ms, rs)
    assert sk.DecryptBatch(ca) == ms, "alt round trip"
    s = pk.AddBatch(c1, ca)
    assert sk.DecryptBatch(s) == [(2 * m) % n for m in ms], "add"
    assert pk.SubBatch(s, ca) == c1, "sub inverts add"
    mod2 = pa.Modulus(ctx, n2)
    inv = mod2.inv_batch(c1)
    assert mod2.mul_batch(inv, c1) == [1] * B, "x * x^-1"
    ks = [rng.randrange(n) for _ in range(B)]
    assert sk.DecryptBatch(pk.ConstMultBatch(c1, ks)) == [m * kk % n for m, kk in zip(ms, ks)], "per-ciphertext ConstMult"
    B2 = B // 4
    m2 = [rng.randrange(n2) for _ in range(B2)]
    c2 = pk.EncryptWithRBatch(m2, rs[:B2], level=1)
    assert sk.DecryptBatch(c2, level=1) == m2, "L2 round trip"
    print(f"{bits}-bit key{' (natural shapes)' if natural else ''}: {B} samples ok ({time.time() - t:.1f}s)", flush=True)
ctx.set_flag("lanes_wanted", 0)
k = K["threshold"]["2048"]
n = int(k["n"], 16); shares = [int(s, 16) for s in k["shares"]]
tk = pa.ThresholdPublicKey(ctx, n, total=5, threshold=3)
ms = [rng.randrange(n) for _ in range(B // 2)]
cts = tk.EncryptBatch(ms)
parts = [tk.PartialDecryptBatch(i, shares[i - 1], cts) for i in (2, 4, 5)]
assert tk.CombinePartialDecryptionsBatch(parts) == ms, "threshold"
print("threshold ok", flush=True)
