#!/usr/bin/env python3
"""One homomorphic operation / share-ZKP call alone, for profiling (rocprofv3 --kernel-trace + tools/trace_gaps.py):
   ops_probe.py add|sub|const|zkp_prove|zkp_verify|l2enc_sk [batch] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
op = sys.argv[1] if len(sys.argv) > 1 else "sub"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (16384 if op.startswith("zkp") or op.startswith("l2") else 65536)
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
KEYS = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
rg = np.random.default_rng(7)
def below(mod, nb, cnt):
    raw = rg.integers(0, 256, size=(cnt, nb), dtype=np.uint8); raw[:, 0] %= np.uint8(max(1, min(255, mod >> (8 * (nb - 1))))); return raw
tb = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
if op.startswith("zkp"):
    kt = KEYS["threshold"]["2048"]
    n, shares, v, vks = int(kt["n"], 16), [int(s, 16) for s in kt["shares"]], int(kt["v"], 16), [int(x, 16) for x in kt["vks"]]
    tk = pa.ThresholdPublicKey(ctx, n, total=5, threshold=3)
    m, r = tb(below(n, 256, B)), below(n, 256, B); r[:, -1] |= 1
    c = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    tk.encrypt_with_r_raw(B, m.data_ptr(), 256, tb(r).data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
    zr = tb(below(n * n, 512, B))
    dec, e, z = (torch.zeros((B, w), dtype=torch.uint8, device=dev) for w in (512, 32, 560))
    prove = lambda: tk.share_zkp_prove_raw(shares[1], v, B, c.data_ptr(), 512, zr.data_ptr(), 512, dec.data_ptr(), 512, e.data_ptr(), z.data_ptr(), 560, MEM_DEVICE)
    ok = np.zeros(B, dtype=np.int32)
    verify = lambda: tk.share_zkp_verify_raw(v, vks[1], B, c.data_ptr(), 512, dec.data_ptr(), 512, e.data_ptr(), z.data_ptr(), 560, ok, MEM_DEVICE)
    prove()
    fn = prove if op == "zkp_prove" else verify
else:
    K = KEYS["paillier"]["2048"]
    p, q = int(K["p"], 16), int(K["q"], 16)
    n = p * q
    pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
    m, r = tb(below(n, 256, B)), below(n, 256, B); r[:, -1] |= 1
    r = tb(r)
    c = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    pk.encrypt_with_r_raw(B, m.data_ptr(), 256, r.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
    c2 = torch.flip(c, dims=[0]).contiguous()
    o = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
    k50 = pow(50, 50, n * n); kb = np.frombuffer(k50.to_bytes(36, "big"), dtype=np.uint8).copy()
    m2 = tb(below(n * n, 512, B))
    fn = {"add": lambda: pk.add_raw(B, c.data_ptr(), 512, c2.data_ptr(), 512, o.data_ptr(), 512, MEM_DEVICE),
          "sub": lambda: pk.sub_raw(B, c.data_ptr(), 512, c2.data_ptr(), 512, o.data_ptr(), 512, MEM_DEVICE),
          "const": lambda: pk.const_mult_raw(B, c.data_ptr(), 512, kb, kb.size, 0, o.data_ptr(), 512, MEM_DEVICE),
          "l2enc_sk": lambda: sk.encrypt_with_r_raw(B, m2.data_ptr(), 512, r.data_ptr(), 256, o.data_ptr(), 768, MEM_DEVICE, level=1),
          "l2enc_pk": lambda: pk.encrypt_with_r_raw(B, m2.data_ptr(), 512, r.data_ptr(), 256, o.data_ptr(), 768, MEM_DEVICE, level=1)}[op]
fn(); fn(); torch.cuda.synchronize()
print("PROBE_BEGIN", flush=True)
for _ in range(REPS):
    t = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(op, B, round((time.perf_counter() - t) * 1e3, 3), "ms", ctx.last_profile(), flush=True)
if op == "zkp_verify":
    assert ok.all()
