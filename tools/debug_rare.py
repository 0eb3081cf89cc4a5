import json, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import paillier_amd as pa
from paillier_amd.api import ints_to_be, be_to_ints
k = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
p, q = int(k["p"], 16), int(k["q"], 16)
n, lam = p*q, (p-1)*(q-1); n2 = n*n
ctx = pa.Context(0); pk = pa.PublicKey(ctx, n, n+1); sk = pa.SecretKey(ctx, pk, lam)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(1234)
pb, cb = 256, 512
def rb(count):
    raw = rng.integers(0, 256, size=(count, pb), dtype=np.uint8); raw[:,0] %= np.uint8(n >> (8*(pb-1))); return raw
m = rb(B); r = rb(B); r[:,-1] |= 1
c = np.zeros((B, cb), np.uint8)
pk.encrypt_with_r_raw(B, m, pb, r, pb, c, cb)
c2 = np.zeros((B, cb), np.uint8)
pk.encrypt_with_r_raw(B, m, pb, r, pb, c2, cb)
print("encrypt deterministic:", (c == c2).all(), "differing lanes:", np.nonzero((c != c2).any(axis=1))[0][:20])
out = np.zeros((B, pb), np.uint8); st = np.zeros(B, np.int32)
sk.decrypt_raw(B, c, cb, out, pb, status=st)
out2 = np.zeros((B, pb), np.uint8)
sk.decrypt_raw(B, c, cb, out2, pb)
print("decrypt deterministic:", (out == out2).all())
bad = np.nonzero((out != m).any(axis=1))[0]
print("bad lanes:", bad[:40], "count", len(bad), "status nonzero:", np.nonzero(st)[0][:20])
mi = be_to_ints(m[bad]); ri = be_to_ints(r[bad]); ci = be_to_ints(c[bad]); oi = be_to_ints(out[bad])
for j, lane in enumerate(bad[:8]):
    c_ref = (1 + mi[j]*n) * pow(ri[j], n, n2) % n2
    enc_ok = c_ref == ci[j]
    # decrypt of the GPU ciphertext by python
    u = pow(ci[j], lam, n2); mref = ((u-1)//n) * pow(lam, -1, n) % n
    print(f"lane {lane}: block {lane//256} tid {lane%256} enc_ok={enc_ok} dec_matches_python_of_gpu_c={mref == oi[j]} m==python_dec={mref==mi[j]}")
    if not enc_ok:
        d = c_ref ^ ci[j]
        print("   enc xor bits:", d.bit_length(), bin(d).count("1"))
    else:
        d = mi[j] ^ oi[j]; print("   dec xor bitlen", d.bit_length(), "popcount", bin(d).count("1"))
