#!/usr/bin/env python3
"""PartialDecrypt-2048 at 16 384 and 32 768 ciphertexts on the default kernel and with EIGHT lanes per number forced
(vm_asm_19_96 at two / four waves per SIMD): the eight-lane slicing is for latency (<= 8 192 numbers: 65 -> 42 ms), not for
throughput -- at 16 384 numbers it takes 69.0 ms against 64.7 ms on four lanes at one wave per SIMD (the per-row overhead is
paid twice; DESIGN.md 2, "Eight lanes per number")."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
kt = K["threshold"]["2048"]
tn, shares = int(kt["n"], 16), [int(s, 16) for s in kt["shares"]]
tk = pa.ThresholdPublicKey(ctx, tn, total=5, threshold=3)
rng = np.random.default_rng(1)
for B in (16384, 32768):
    raw = rng.integers(0, 256, size=(B, 512), dtype=np.uint8); raw[:, 0] = 0
    c = torch.from_numpy(raw).to(dev); o = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    for lw in (0, 8 * B):
        ctx.set_flag("lanes_wanted", lw)
        for _ in range(2):
            t = time.perf_counter(); tk.partial_decrypt_raw(shares[0], B, c.data_ptr(), 512, o.data_ptr(), 512, MEM_DEVICE); dt = time.perf_counter() - t
        print(B, lw, round(dt * 1e3, 2), "ms", ctx.last_profile()["kernel"], flush=True)
    ctx.set_flag("lanes_wanted", 0)
