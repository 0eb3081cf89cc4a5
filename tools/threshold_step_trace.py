#!/usr/bin/env python3
"""One threshold step of bench config 4 on one GPU (pgpu_partial_decrypt_units over all three shares + pgpu_combine), for a kernel trace:
where the ~4 ms outside the ladder go.   rocprofv3 --kernel-trace ... -- python3 tools/threshold_step_trace.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["threshold"]["2048"]
n, shares = int(K["n"], 16), [int(s, 16) for s in K["shares"]]
ids = [1, 3, 5]
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
tk = pa.ThresholdPublicKey(ctx, n, total=5, threshold=3)
B = 16384
rg = np.random.default_rng(4)
def below(cnt):
    raw = rg.integers(0, 256, size=(cnt, 256), dtype=np.uint8); raw[:, 0] %= np.uint8(max(1, min(255, n >> (8 * 255)))); return raw
m, r = below(B), below(B); r[:, -1] |= 1
tm, tr = torch.from_numpy(m).to(dev), torch.from_numpy(r).to(dev)
c = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
tk.encrypt_with_r_raw(B, tm.data_ptr(), 256, tr.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
sh = [shares[i - 1] for i in ids]
parts = torch.empty((3 * B, 512), dtype=torch.uint8, device=dev)
out = torch.empty((B, 256), dtype=torch.uint8, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    tk.partial_decrypt_units_raw(sh, B, c.data_ptr(), 512, 0, 3 * B, parts.data_ptr(), 512, MEM_DEVICE)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    p3 = parts.view(3, B, 512)
    tk.combine_raw(ids, B, [p3[s].data_ptr() for s in range(3)], 512, out.data_ptr(), 256, MEM_DEVICE)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"partial {1e3 * (t1 - t):.2f} ms  combine {1e3 * (t2 - t1):.2f} ms  ok {bool(torch.equal(out, tm))}", flush=True)
