#!/usr/bin/env python3
"""Decrypt-2048 and Encrypt-2048 at several batch sizes with the pair kernels on (automatic choice) and off."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
k = K["paillier"]["2048"]; p, q = int(k["p"], 16), int(k["q"], 16); n = p * q
pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
rng = np.random.default_rng(1)
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0
for B in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
    c = torch.from_numpy(rng.integers(1, 255, size=(B, 512), dtype=np.uint8)).to(dev); c[:, 0] = 0
    m = torch.from_numpy(rng.integers(0, 255, size=(B, 256), dtype=np.uint8)).to(dev)
    out = torch.zeros((B, 256), dtype=torch.uint8, device=dev); co = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    res = {}
    for pair in (1, 0):
        ctx.set_flag("pair", pair)
        d = t(lambda: sk.decrypt_raw(B, c.data_ptr(), 512, out.data_ptr(), 256, MEM_DEVICE)); o1 = out.clone()
        e = t(lambda: pk.encrypt_with_r_raw(B, m.data_ptr(), 256, m.data_ptr(), 256, co.data_ptr(), 512, MEM_DEVICE)); o2 = co.clone()
        res[pair] = (B / d, B / e, o1, o2)
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    print(f"B={B}: decrypt pair {res[1][0]:.0f}/s  off {res[0][0]:.0f}/s   encrypt pair {res[1][1]:.0f}/s  off {res[0][1]:.0f}/s", flush=True)
ctx.set_flag("pair", 1)
