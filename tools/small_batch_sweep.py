#!/usr/bin/env python3
"""Per-ciphertext cost of PartialDecrypt-2048 and Decrypt-2048 as the batch shrinks (what one of 8 GPUs
sees of BASELINE config 4 is 2048 ciphertexts).  Prints one JSON line per (op, batch)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE

K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
kt = K["threshold"]["2048"]
tn, shares = int(kt["n"], 16), [int(s, 16) for s in kt["shares"]]
tk = pa.ThresholdPublicKey(ctx, tn, total=5, threshold=3)
kp = K["paillier"]["2048"]
p, q = int(kp["p"], 16), int(kp["q"], 16)
pk = pa.PublicKey(ctx, p * q)
sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
rng = np.random.default_rng(1)
BMAX = 65536
raw = rng.integers(0, 256, size=(BMAX, 512), dtype=np.uint8)
raw[:, 0] = 0
c = torch.from_numpy(raw).to(dev)
o5 = torch.zeros((BMAX, 512), dtype=torch.uint8, device=dev)
o2 = torch.zeros((BMAX, 256), dtype=torch.uint8, device=dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


for B in [int(x) for x in (sys.argv[1:] or [512, 1024, 2048, 4096, 8192, 16384, 32768, 65536])]:
    dt = timed(lambda: tk.partial_decrypt_raw(shares[0], B, c.data_ptr(), 512, o5.data_ptr(), 512, MEM_DEVICE))
    pr = ctx.last_profile()
    print(json.dumps({"op": "partial_decrypt_2048", "batch": B, "ms": dt * 1e3, "us_per_ct": dt / B * 1e6, "per_s": B / dt,
                      "kernel": pr["kernel"], "vm_ms": pr["vm_ms"]}), flush=True)
    dt = timed(lambda: sk.decrypt_raw(B, c.data_ptr(), 512, o2.data_ptr(), 256, MEM_DEVICE))
    pr = ctx.last_profile()
    print(json.dumps({"op": "decrypt_2048", "batch": B, "ms": dt * 1e3, "us_per_ct": dt / B * 1e6, "per_s": B / dt,
                      "kernel": pr["kernel"], "vm_ms": pr["vm_ms"]}), flush=True)


# ---- the same small batches overlapped: `width` independent PartialDecrypt calls (e.g. the servers of a threshold
# decryption) on contexts with streams of their own, driven from host threads (paillier_amd.concurrent.Lanes)
from paillier_amd.concurrent import Lanes
# "spread" (plan::lds_share): a main-stream ladder of at most one workgroup per CU asks for just over half a CU's LDS; with a CU partition
# narrower than the launch the rule switches itself off -- both settings are measured
for width, part, spread in ((3, False, 1), (3, False, 0), (3, True, 1), (8, True, 1), (8, True, 0), (8, False, 1), (8, False, 0)):
    lanes = Lanes(0, width, partition_cus=part)
    for cx in lanes.contexts:
        cx.set_flag("spread", spread)
    bufs = [torch.zeros((4096, 512), dtype=torch.uint8, device=dev) for _ in range(width)]

    def call(cx, st, item):
        k, B = item
        if "tk" not in st:
            st["tk"] = pa.ThresholdPublicKey(cx, tn, total=5, threshold=3)
        st["tk"].partial_decrypt_raw(shares[k % 5], B, c.data_ptr(), 512, bufs[k].data_ptr(), 512, MEM_DEVICE)

    for B in (1024, 2048, 4096):
        items = [(k, B) for k in range(width)]
        lanes.map(call, items)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            lanes.map(call, items)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        print(json.dumps({"op": "partial_decrypt_2048_overlapped", "calls_in_flight": width, "cu_partition": part, "spread": spread, "batch_per_call": B, "ms": dt * 1e3,
                          "us_per_ct": dt / (B * width) * 1e6, "per_s": B * width / dt}), flush=True)
    lanes.close()

# ---- the units of three servers in ONE launch (pgpu_partial_decrypt_indexed: per-unit exponents)
for B in (2048, 4096, 16384):
    S = 3
    rows = c[:B].repeat(S, 1).contiguous()
    idx = np.repeat(np.arange(S, dtype=np.int32), B)
    o = torch.zeros((S * B, 512), dtype=torch.uint8, device=dev)
    dt = timed(lambda: tk.partial_decrypt_indexed_raw(shares[:S], idx, S * B, rows.data_ptr(), 512, o.data_ptr(), 512, MEM_DEVICE))
    pr = ctx.last_profile()
    print(json.dumps({"op": "partial_decrypt_2048_indexed", "servers": S, "ciphertexts": B, "ms": dt * 1e3,
                      "us_per_unit": dt / (S * B) * 1e6, "units_per_s": S * B / dt, "kernel": pr["kernel"]}), flush=True)
