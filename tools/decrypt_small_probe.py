#!/usr/bin/env python3
"""Decrypt-2048 of small batches: the CRT ladders on the eight-lane pair kernel (flag lanes8 1, plan::crt_pair_lanes8) against the
two-lane kernel (lanes8 0).  One JSON line per batch size: call ms (best of 5), the ladder launch's ms and kernel."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
p, q = int(K["p"], 16), int(K["q"], 16)
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
pk = pa.PublicKey(ctx, p * q); sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
rng = np.random.default_rng(1)
for B in [int(a) for a in sys.argv[1:]] or (256, 1024, 2048, 4096, 8192):
    raw = rng.integers(0, 256, size=(B, 512), dtype=np.uint8); raw[:, 0] = 0
    c = torch.from_numpy(raw).to(dev); o = torch.zeros((B, 256), dtype=torch.uint8, device=dev)
    row = {"batch": B}
    for l8 in (1, 0):
        ctx.set_flag("lanes8", l8)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); sk.decrypt_raw(B, c.data_ptr(), 512, o.data_ptr(), 256, MEM_DEVICE); best = min(best, time.perf_counter() - t)
        pr = ctx.last_profile()
        row["lanes8" if l8 else "two_lanes"] = [round(best * 1e3, 2), round(pr["vm_ms"], 2), pr["kernel"]]
    ctx.set_flag("lanes8", 1)
    print(json.dumps(row), flush=True)
