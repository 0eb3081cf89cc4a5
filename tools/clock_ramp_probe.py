#!/usr/bin/env python3
"""Does the headline kernel pay a clock ramp after the idle gap between two blocking calls?  Times the VM kernel of a
65536-ciphertext Decrypt-2048 (HIP events around the kernel only) with and without a few ms of unrelated GPU work queued on
the same stream right before the call."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
p, q = int(K["p"], 16), int(K["q"], 16)
n = p * q
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
B = 65536
rg = np.random.default_rng(1)
raw = rg.integers(0, 256, size=(B, 512), dtype=np.uint8); raw[:, 0] = 0
c = torch.from_numpy(raw).to(dev); out = torch.zeros((B, 256), dtype=torch.uint8, device=dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
def run(burn_iters, sleep_ms):
    ms = []
    for _ in range(6):
        torch.cuda.synchronize()
        if sleep_ms: time.sleep(sleep_ms * 1e-3)
        for _ in range(burn_iters):
            (a @ a)
        sk.decrypt_raw(B, c.data_ptr(), 512, out.data_ptr(), 256, MEM_DEVICE)
        ms.append(ctx.last_profile()["vm_ms"])
    return [round(x, 2) for x in ms]
print("no burn, no sleep   ", run(0, 0))
print("no burn, 50 ms idle ", run(0, 50))
print("burn x1 (~1 ms)     ", run(1, 0))
print("burn x5             ", run(5, 0))
print("burn x20            ", run(20, 0))
print("burn x20, idle 50 ms", run(20, 50))
