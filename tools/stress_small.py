#!/usr/bin/env python3
"""Many samples through the LATENCY kernels of round 5 (vm_asm_10_4, 10_96, 19_112, 10_128, 19_160), which only small
batches reach -- rare data-dependent bugs (a lazy value landing in [N, 2N), a carry out of a slice) show at ~1e-4 rates, and the unit
tests use a few hundred numbers.  Size-independent properties on a 2048-bit key, `rounds` batches of `B` numbers each:
Decrypt(Encrypt(m)) == m at both levels, the key holder's Encrypt == the public one, NestedRandomize -> prove -> verify.
   stress_small.py [rounds] [B]"""
import json, os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import paillier_amd as pa
from paillier_amd import ENC_LEVEL_TWO

K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ctx = pa.Context(0)
p, q = int(K["p"], 16), int(K["q"], 16)
n, lam = p * q, (p - 1) * (q - 1)
n2 = n * n
pk = pa.PublicKey(ctx, n, n + 1)
sk = pa.SecretKey(ctx, pk, lam)
rng = random.Random(2025)
seen = set()
t0 = time.time()
for it in range(rounds):
    ms = [rng.randrange(n) for _ in range(B)]
    rs = [rng.randrange(1, n) for _ in range(B)]
    c1 = pk.EncryptWithRBatch(ms, rs)                                  # r^n modulo n^2: sixteen lanes per number (B <= 2 048)
    seen.add(ctx.last_profile()["kernel"])
    assert sk.EncryptWithRBatch(ms, rs) == c1, "key holder's Encrypt"  # through the primes (four lanes) and p^2, q^2 (eight lanes)
    assert sk.DecryptBatch(c1) == ms, "L1 round trip"                  # CRT ladders on eight lanes
    seen.add(ctx.last_profile()["kernel"])
    c2 = sk.EncryptWithRBatch(c1, rs, ENC_LEVEL_TWO)                   # the lift modulo p^3, q^3: two lanes per digit
    assert sk.DecryptBatch(c2, level=ENC_LEVEL_TWO) == c1, "L2 round trip"
    seen.add(ctx.last_profile()["kernel"])
    if it % 4 == 0:
        S = min(B, 1024)
        a_s, b_s = [rng.randrange(1, n) for _ in range(S)], [rng.randrange(1, n) for _ in range(S)]
        ct2 = pk.NestedRandomizeWithABBatch(c2[:S], a_s, b_s)          # x^(e0) W^n modulo n^3: four lanes per digit
        seen.add(ctx.last_profile()["kernel"])
        xs, ys = [rng.randrange(1, n) for _ in range(S)], [rng.randrange(1, n) for _ in range(S)]
        al, es, fs = sk.ProveDDLEQInstancesBatch(c2[:S], ct2, a_s, b_s, xs, ys)
        assert all(pk.VerifyDDLEQInstancesBatch(c2[:S], ct2, xs, ys, al, es, fs)), "prove -> verify"
    print(f"round {it + 1}/{rounds} ok ({time.time() - t0:.0f} s)", flush=True)
print("kernels seen as the dominant one of a call:", sorted(seen))
print(f"all properties hold on {rounds} x {B} numbers")
