#!/usr/bin/env python3
"""Small-batch latency of the key holder's calls on a 2048-bit key: Decrypt (CRT ladders on the eight-lane pair kernel vm_asm_10_96
against the two-lane kernel), level-two Decrypt (ladders modulo p^3, q^3 with two lanes per digit, vm_asm_19_112, against one) --
flag lanes8 1 | 0 -- and the public Encrypt (r^n modulo n^2 on sixteen lanes, vm_asm_10_128, against eight: flag lanes16).  One JSON
line per call and batch size: call ms (best of 5), the ladder launch's ms and kernel.   decrypt_small_probe.py [batch ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
from paillier_amd import ENC_LEVEL_TWO
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
p, q = int(K["p"], 16), int(K["q"], 16)
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
pk = pa.PublicKey(ctx, p * q); sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
rng = np.random.default_rng(1)
def best(fn):
    b = 1e9
    for _ in range(5):
        t = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t)
    pr = ctx.last_profile()
    return [round(b * 1e3, 2), round(pr["vm_ms"], 2), pr["kernel"]]
for B in [int(a) for a in sys.argv[1:]] or (256, 1024, 2048, 4096, 8192):
    raw = rng.integers(0, 256, size=(B, 768), dtype=np.uint8); raw[:, 0] = 0
    c1 = torch.from_numpy(np.ascontiguousarray(raw[:, :512])).to(dev); c2 = torch.from_numpy(raw).to(dev)
    o1 = torch.zeros((B, 256), dtype=torch.uint8, device=dev); o2 = torch.zeros((B, 512), dtype=torch.uint8, device=dev)
    m = torch.from_numpy(np.ascontiguousarray(raw[:, 512:])).to(dev); r = torch.from_numpy(np.ascontiguousarray(raw[:, 256:512] | 1)).to(dev)
    rows = {"decrypt": {}, "decrypt_l2": {}, "encrypt": {}, "encrypt_l2": {}}
    for flag, on in (("lanes8", 1), ("lanes8", 0)):
        ctx.set_flag(flag, on)
        rows["decrypt"]["wide" if on else "narrow"] = best(lambda: sk.decrypt_raw(B, c1.data_ptr(), 512, o1.data_ptr(), 256, MEM_DEVICE))
        rows["decrypt_l2"]["wide" if on else "narrow"] = best(lambda: sk.decrypt_raw(B, c2.data_ptr(), 768, o2.data_ptr(), 512, MEM_DEVICE, level=ENC_LEVEL_TWO))
    ctx.set_flag("lanes8", 1)
    for on in (1, 0):
        ctx.set_flag("lanes16", on)
        rows["encrypt"]["wide" if on else "narrow"] = best(lambda: pk.encrypt_with_r_raw(B, m.data_ptr(), 256, r.data_ptr(), 256, c1.data_ptr(), 512, MEM_DEVICE))
        # level two: (r^n mod n^2)^n mod n^3 -- sixteen lanes, then four lanes per digit (vm_asm_19_160) against eight and two (vm_asm_37_112)
        rows["encrypt_l2"]["wide" if on else "narrow"] = best(lambda: pk.encrypt_with_r_raw(B, c1.data_ptr(), 512, r.data_ptr(), 256, c2.data_ptr(), 768, MEM_DEVICE, level=ENC_LEVEL_TWO))
    ctx.set_flag("lanes16", 1)
    for k, v in rows.items():
        print(json.dumps({"call": k, "batch": B, **v}), flush=True)
