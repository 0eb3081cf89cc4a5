#!/usr/bin/env python3
"""Encrypt-2048 (r^n mod n^2, 148 limbs) at several batch sizes with the re-slicing threshold forced both ways:
which of the wave-sliced (74,2) kernel and the 4-lane (37,4) kernel wins below full occupancy."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
k = K["paillier"]["2048"]; n = int(k["p"], 16) * int(k["q"], 16)
pk = pa.PublicKey(ctx, n)
rng = np.random.default_rng(1)
for B in (4096, 8192, 16384, 32768, 65536):
    m = torch.from_numpy(rng.integers(0, 255, size=(B, 256), dtype=np.uint8)).to(dev)
    r = torch.from_numpy(rng.integers(1, 255, size=(B, 256), dtype=np.uint8)).to(dev)
    c = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    outs = []
    for lanes in (1, 1 << 30):      # 1: natural shape (74,2 wave-sliced); huge: always re-slice to (37,4)
        ctx.set_flag("lanes_wanted", lanes)
        pk.encrypt_with_r_raw(B, m.data_ptr(), 256, r.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
        torch.cuda.synchronize(); t = time.perf_counter()
        pk.encrypt_with_r_raw(B, m.data_ptr(), 256, r.data_ptr(), 256, c.data_ptr(), 512, MEM_DEVICE)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        outs.append((B / dt, c.clone()))
    assert torch.equal(outs[0][1], outs[1][1])
    print(f"B={B}: (74,2)w {outs[0][0]:.0f}/s   (37,4) {outs[1][0]:.0f}/s", flush=True)

# Decrypt-2048 (CRT, two 74-limb segments): (74,1) against the 2-lane (37,2) and 4-lane slicings below full occupancy
lam = (int(k["p"], 16) - 1) * (int(k["q"], 16) - 1)
sk = pa.SecretKey(ctx, pk, lam)
for B in (2048, 4096, 8192, 16384, 32768, 65536):
    c = torch.from_numpy(rng.integers(1, 255, size=(B, 512), dtype=np.uint8)).to(dev); c[:, 0] = 0
    out = torch.zeros((B, 256), dtype=torch.uint8, device=dev)
    res = []
    for lanes in (1, 2 * B * 2, 1 << 30):      # natural (74,1); one re-slice (37,2); as many as possible (37,2 is the limit for 74)
        ctx.set_flag("lanes_wanted", lanes)
        sk.decrypt_raw(B, c.data_ptr(), 512, out.data_ptr(), 256, MEM_DEVICE)
        torch.cuda.synchronize(); t = time.perf_counter()
        sk.decrypt_raw(B, c.data_ptr(), 512, out.data_ptr(), 256, MEM_DEVICE)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        res.append((B / dt, out.clone()))
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][1], res[2][1])
    print(f"B={B}: (74,1) {res[0][0]:.0f}/s   one re-slice {res[1][0]:.0f}/s   max re-slice {res[2][0]:.0f}/s", flush=True)
