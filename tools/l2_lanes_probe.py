#!/usr/bin/env python3
"""Level-two EncryptWithR at 2048 bits, small to large batches, with the two-lanes-per-digit three-digit kernel (vm_asm_37_112:
eight lanes per number) allowed and not (flag "lanes8"): where the wider slicing pays (launches below one wave per SIMD) and
that both give the same bytes."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
p, q = int(K["p"], 16), int(K["q"], 16); n = p * q
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
pk = pa.PublicKey(ctx, n)
rng = np.random.default_rng(3)
for B in (64, 256, 1024, 2048, 4096, 8192, 16384):
    m = rng.integers(0, 256, size=(B, 512), dtype=np.uint8); m[:, 0] = 0
    r = rng.integers(0, 256, size=(B, 256), dtype=np.uint8); r[:, 0] = 0; r[:, -1] |= 1
    dm, dr = torch.from_numpy(m).to(dev), torch.from_numpy(r).to(dev)
    outs = []
    for l8 in (1, 0):
        ctx.set_flag("lanes8", l8)
        o = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); pk.encrypt_with_r_raw(B, dm.data_ptr(), 512, dr.data_ptr(), 256, o.data_ptr(), 768, MEM_DEVICE, level=1)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        prof = ctx.last_profile()
        outs.append(o.cpu())
        print(json.dumps({"batch": B, "lanes8": l8, "ms": round(best * 1e3, 3), "kernel": prof["kernel"], "vm_ms": round(prof.get("vm_ms", 0), 3)}), flush=True)
    assert torch.equal(outs[0], outs[1]), B
    ctx.set_flag("lanes8", 1)
print("same bytes on both kernels")
