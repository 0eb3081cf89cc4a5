#!/usr/bin/env python3
"""A/B of the wave priorities on whatever box this runs on: Decrypt-2048 x 65536 and Encrypt-2048 x 65536 with
pgpu_ctx_set_flag("fair", 1 / 0), kernel times from HIP events, plus the box's clocks under load."""
import json, os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
p, q = int(K["p"], 16), int(K["q"], 16); n = p * q
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
B = 65536
rg = np.random.default_rng(1)
m = rg.integers(0, 256, (B, 256), dtype=np.uint8); m[:, 0] = 0
r = rg.integers(0, 256, (B, 256), dtype=np.uint8); r[:, 0] = 0; r[:, -1] |= 1
md, rd = torch.from_numpy(m).to(dev), torch.from_numpy(r).to(dev)
c = torch.zeros((B, 512), dtype=torch.uint8, device=dev); o = torch.zeros((B, 256), dtype=torch.uint8, device=dev)
pk.encrypt_with_r_raw(B, md.data_ptr(), 256, rd.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
res = {}
for rep in range(2):
    for fair in (1, 0):
        ctx.set_flag("fair", fair)
        ms = []
        for _ in range(4):
            sk.decrypt_raw(B, c.data_ptr(), 512, o.data_ptr(), 256, MEM_DEVICE); ms.append(ctx.last_profile()["vm_ms"])
        me = []
        for _ in range(2):
            pk.encrypt_with_r_raw(B, md.data_ptr(), 256, rd.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE); me.append(ctx.last_profile()["vm_ms"])
        res[(rep, fair)] = (round(min(ms), 2), round(min(me), 2))
        print("rep", rep, "fair", fair, "decrypt kernel ms", [round(x, 2) for x in ms], "encrypt kernel ms", [round(x, 2) for x in me], flush=True)
ctx.set_flag("fair", 1)
assert torch.equal(o, md)
