#!/usr/bin/env python3
"""What one rank of the sharded threshold flow (paillier_amd/dist.py) computes at N = 1, 2, 4, 8 for config 4 (16384 ciphertexts,
t = 3): its (server, ciphertext) units through pgpu_partial_decrypt_indexed, against the alternatives on the same units."""
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
raw = rg.integers(0, 256, size=(B, 512), dtype=np.uint8); raw[:, 0] = 0
c = torch.from_numpy(raw).to(dev)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
sh = [shares[i - 1] for i in ids]
for world in (1, 2, 4, 8):
    units = 3 * B // world
    us = np.arange(0, units, dtype=np.int64)              # rank 0's range, server-major
    rows = c[torch.from_numpy(us % B).to(dev)].contiguous()
    out = torch.empty((units, 512), dtype=torch.uint8, device=dev)
    t_rng = timed(lambda: tk.partial_decrypt_units_raw(sh, B, c.data_ptr(), 512, 0, units, out.data_ptr(), 512, MEM_DEVICE))
    k_rng = ctx.last_profile()["kernel"]
    # the LAST rank's range as well (at N = 2: the second half of server 1 + server 2 whole)
    t_rng_last = timed(lambda: tk.partial_decrypt_units_raw(sh, B, c.data_ptr(), 512, 3 * B - units, 3 * B, out.data_ptr(), 512, MEM_DEVICE))
    t_idx = timed(lambda: tk.partial_decrypt_indexed_raw(sh, (us // B).astype(np.int32), units, rows.data_ptr(), 512, out.data_ptr(), 512, MEM_DEVICE))
    # the same units server by server (what partial_fn would do)
    def per_server():
        u = 0
        while u < units:
            s, i0 = divmod(u, B); cnt = min(units - u, B - i0)
            tk.partial_decrypt_raw(sh[s], cnt, c[i0:i0 + cnt].data_ptr(), 512, out[u:u + cnt].data_ptr(), 512, MEM_DEVICE)
            u += cnt
    t_srv = timed(per_server)
    # ciphertext-sharded instead: B / world ciphertexts under all three shares, one chain of squarings
    cnt = B // world
    outs = [torch.empty((cnt, 512), dtype=torch.uint8, device=dev) for _ in ids]
    t_multi = timed(lambda: tk.partial_decrypt_multi_raw(sh, cnt, c[:cnt].data_ptr(), 512, [o.data_ptr() for o in outs], 512, MEM_DEVICE))
    # ciphertext-major shard: the rank's B / world ciphertexts under ALL shares through the units entry point (the whole unit range of
    # the slice: one shared chain of squarings, the lane count the plan picks for so small a batch)
    outc = torch.empty((3 * cnt, 512), dtype=torch.uint8, device=dev)
    t_cm = timed(lambda: tk.partial_decrypt_units_raw(sh, cnt, c[:cnt].data_ptr(), 512, 0, 3 * cnt, outc.data_ptr(), 512, MEM_DEVICE))
    k_cm = ctx.last_profile()["kernel"]
    print(json.dumps({"world": world, "ciphertext_major_units_ms": round(t_cm, 1), "ciphertext_major_kernel": k_cm, "units_per_rank": units, "units_range_ms": round(t_rng, 1), "units_range_last_rank_ms": round(t_rng_last, 1),
                      "units_range_kernel": k_rng, "indexed_ms": round(t_idx, 1), "server_by_server_ms": round(t_srv, 1),
                      "ciphertext_sharded_multi_ms": round(t_multi, 1), "kernel": ctx.last_profile()["kernel"]}), flush=True)
