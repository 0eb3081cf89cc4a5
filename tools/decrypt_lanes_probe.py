#!/usr/bin/env python3
"""Decrypt-2048 between half a wave and one wave per SIMD (16 384 < batch < 32 768): the default choice (two lanes per number
below 32 768) against one lane forced (lanes_wanted = 1)."""
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
for B in (12288, 16384, 20480, 24576, 28672, 30720, 32768):
    raw = rng.integers(0, 256, size=(B, 512), dtype=np.uint8); raw[:, 0] = 0
    c = torch.from_numpy(raw).to(dev); o = torch.zeros((B, 256), dtype=torch.uint8, device=dev)
    row = {"batch": B}
    for lw in (0, 1):
        ctx.set_flag("lanes_wanted", lw)
        for _ in range(3):
            t = time.perf_counter(); sk.decrypt_raw(B, c.data_ptr(), 512, o.data_ptr(), 256, MEM_DEVICE); dt = time.perf_counter() - t
        row["default" if lw == 0 else "one_lane"] = [round(dt * 1e3, 2), ctx.last_profile()["kernel"]]
    ctx.set_flag("lanes_wanted", 0)
    print(json.dumps(row), flush=True)
