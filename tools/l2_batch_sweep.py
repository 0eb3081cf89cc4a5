#!/usr/bin/env python3
"""Level-two EncryptWithR throughput against batch size (one MI355X): 16384 numbers = one wave per SIMD of the
three-digit kernel (4 lanes per number), 32768 = two.  Prints one JSON line per batch."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import paillier_amd.api as pa

KEYS = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "keys.json")))
k = KEYS["paillier"]["2048"]
p, q = int(k["p"], 16), int(k["q"], 16)
n = p * q
ctx = pa.Context()
pk = pa.PublicKey(ctx, n, n + 1)
sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
dev = torch.device("cuda:0")
rg = np.random.default_rng(1)
for B in (int(a) for a in (sys.argv[1:] or ["8192", "16384", "32768", "65536"])):
    m = rg.integers(0, 256, (B, 512), dtype=np.uint8); m[:, 0] = 0
    r = rg.integers(0, 256, (B, 256), dtype=np.uint8); r[:, 0] = 0; r[:, -1] |= 1
    md, rd = torch.from_numpy(m).to(dev), torch.from_numpy(r).to(dev)
    c = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
    o = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    f = lambda: pk.encrypt_with_r_raw(B, md.data_ptr(), 512, rd.data_ptr(), 256, c.data_ptr(), 768, pa.MEM_DEVICE, level=1)
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    pr = ctx.last_profile()
    sk.decrypt_raw(B, c.data_ptr(), 768, o.data_ptr(), 512, pa.MEM_DEVICE, level=1)
    print(json.dumps({"batch": B, "ms": dt * 1e3, "per_s": B / dt, "vm_ms": pr["vm_ms"], "kernel": pr["kernel"],
                      "frac": pr["vm_mads"] / (pr["vm_ms"] * 1e-3) / 39.3216e12, "round_trip": bool(torch.equal(o, md))}), flush=True)
