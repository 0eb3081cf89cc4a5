import json, os, sys, random, time
sys.path.insert(0, '/root/repo')
import paillier_amd as pa
K = json.load(open("tests/golden/keys.json"))
ctx = pa.Context(0)
k = K["paillier"]["2048"]
p, q = int(k["p"], 16), int(k["q"], 16)
n = p*q; lam=(p-1)*(q-1)
pk = pa.PublicKey(ctx, n, n + 1); sk = pa.SecretKey(ctx, pk, lam)
rng = random.Random(1)
B = 8192
ms = [rng.randrange(n*n) for _ in range(B)]
rs = [rng.randrange(1, n) for _ in range(B)]
cts = pk.EncryptWithRBatch(ms, rs, level=1)
for fl in (0, 1):
    for it in range(2):
        t = time.time(); out = sk.DecryptBatch(cts, level=1, flags=fl); dt = time.time() - t
    assert out == ms
    print("flags", fl, "L2 decrypt/s (incl. host packing)", B/dt)
