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
# round 3: small batches (the eight-lane pair kernel, vm_asm_19_96), a rank's unit range with and without the shared chain,
# DDLEQ at secpar 8 with library-drawn randomness, all against size-independent properties
import numpy as np
from paillier_amd import protocols as pr
from paillier_amd.api import be_to_ints, ints_to_be
Bs = min(B, 6000)
cts_s = cts[:Bs]
for i in (1, 3):
    one = tk.PartialDecryptBatch(i, shares[i - 1], cts_s)
    ctx.set_flag("lanes8", 0)
    other = tk.PartialDecryptBatch(i, shares[i - 1], cts_s)
    ctx.set_flag("lanes8", 1)
    assert one == other, "eight-lane pair kernel == four-lane"
cb = tk.cipher_bytes()
rows = ints_to_be(cts_s, cb)
sh3 = [shares[1], shares[3], shares[4]]
for ub, ue in ((0, 3 * Bs), (Bs // 2, 2 * Bs + 17), (Bs + 5, 3 * Bs - 9)):
    out = np.zeros((ue - ub, cb), dtype=np.uint8)
    tk.partial_decrypt_units_raw(sh3, Bs, rows, cb, ub, ue, out, cb)
    got = be_to_ints(out)
    ctx.set_flag("shared_chain", 0)
    out2 = np.zeros((ue - ub, cb), dtype=np.uint8)
    tk.partial_decrypt_units_raw(sh3, Bs, rows, cb, ub, ue, out2, cb)
    ctx.set_flag("shared_chain", 1)
    assert got == be_to_ints(out2), "unit range: shared chain == separate ladders"
full = np.zeros((3 * Bs, cb), dtype=np.uint8)
tk.partial_decrypt_units_raw(sh3, Bs, rows, cb, 0, 3 * Bs, full, cb)
cols = [be_to_ints(full[s * Bs:(s + 1) * Bs]) for s in range(3)]
assert tk.CombinePartialDecryptionsBatch([(2, cols[0]), (4, cols[1]), (5, cols[2])]) == ms[:Bs], "threshold through the unit range"
print("units / eight lanes ok", flush=True)
k = K["paillier"]["2048"]
p, q = int(k["p"], 16), int(k["q"], 16)
n = p * q
pk = pa.PublicKey(ctx, n, n + 1); sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
S = max(8, min(B // 40, 256))
c1 = pk.NestedEncryptBatch([rng.randrange(n) for _ in range(S)])
a_s, b_s = pk.random_units(S), pk.random_units(S)
c2 = pr.nested_randomize_with_ab_batch(pk, c1, a_s, b_s)
proofs = pr.prove_ddleq_batch(sk, 8, c1, c2, a_s, b_s)
assert pr.verify_ddleq_proof_batch(pk, c1, c2, proofs) == [True] * S, "ProveDDLEQ(secpar 8) verifies"
assert sk.NestedDecryptBatch(c2) == sk.NestedDecryptBatch(c1), "NestedRandomize keeps the plaintext"
print(f"ddleq secpar 8 x {S} statements ok", flush=True)
