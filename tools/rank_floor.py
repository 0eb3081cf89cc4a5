#!/usr/bin/env python3
"""What ONE rank of the strong-scaled bench entries computes at N = 1, 2, 4, 8 (16384 / N statements or ciphertexts), measured on one GPU:
NestedRandomize, DDLEQ prove, DDLEQ verify (config 5) and both threshold shards (config 4: `threshold_2048` = the slowest rank's unit range of
the exchange flow, `threshold_2048_replicated` = a ciphertext slice under all three shares).  Writes the table bench.py reads
(profiles/r05_rank_floor.json) to stdout.   rank_floor.py [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import paillier_amd as pa
from paillier_amd.api import MEM_DEVICE
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
KEYS = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))
K = KEYS["paillier"]["2048"]
p, q = int(K["p"], 16), int(K["q"], 16)
n = p * q
dev = torch.device("cuda", 0)
ctx = pa.Context(0, torch.cuda.current_stream().cuda_stream)
pk = pa.PublicKey(ctx, n); sk = pa.SecretKey(ctx, pk, (p - 1) * (q - 1))
rg = np.random.default_rng(5)
def below(mod, nb, cnt):
    raw = rg.integers(0, 256, size=(cnt, nb), dtype=np.uint8); raw[:, 0] %= np.uint8(max(1, min(255, mod >> (8 * (nb - 1))))); return raw
def unit(cnt):
    a = below(n, 256, cnt); a[:, -1] |= 1; return a
tb = lambda a: torch.from_numpy(a).to(dev)
def best(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return round(min(ts), 1)
out = {"nested_randomize_2048": {}, "ddleq_prove_2048": {}, "ddleq_verify_2048": {}, "threshold_2048": {}, "threshold_2048_replicated": {}}
for world in (1, 2, 4, 8):
    B = 16384 // world
    msg, r1, r2, a_, b_, x_, y_ = below(n, 256, B), unit(B), unit(B), unit(B), unit(B), unit(B), unit(B)
    inner = torch.zeros((B, 512), dtype=torch.uint8, device=dev); ct1 = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
    pk.encrypt_with_r_raw(B, tb(msg).data_ptr(), 256, tb(r1).data_ptr(), 256, inner.data_ptr(), 512, MEM_DEVICE)
    pk.encrypt_with_r_raw(B, inner.data_ptr(), 512, tb(r2).data_ptr(), 256, ct1.data_ptr(), 768, MEM_DEVICE, level=1)
    da, db, dx, dy = tb(a_), tb(b_), tb(x_), tb(y_)
    ct2 = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
    out["nested_randomize_2048"][str(world)] = best(lambda: pk.nested_randomize_with_ab_raw(B, ct1.data_ptr(), da.data_ptr(), db.data_ptr(), ct2.data_ptr(), MEM_DEVICE))
    al = torch.zeros((B, 768), dtype=torch.uint8, device=dev); pe = torch.zeros((B, 512), dtype=torch.uint8, device=dev); pf = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
    out["ddleq_prove_2048"][str(world)] = best(lambda: sk.ddleq_prove_raw(B, ct1.data_ptr(), ct2.data_ptr(), da.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr(), al.data_ptr(), pe.data_ptr(), pf.data_ptr(), MEM_DEVICE))
    ok = np.zeros(B, dtype=np.int32)
    out["ddleq_verify_2048"][str(world)] = best(lambda: pk.ddleq_verify_raw(B, ct1.data_ptr(), ct2.data_ptr(), dx.data_ptr(), dy.data_ptr(), al.data_ptr(), pe.data_ptr(), pf.data_ptr(), ok, MEM_DEVICE))
    assert ok.all(), "a proof was rejected"
    del inner, ct1, ct2, da, db, dx, dy, al, pe, pf
kt = KEYS["threshold"]["2048"]
tn, shares = int(kt["n"], 16), [int(s, 16) for s in kt["shares"]]
tk = pa.ThresholdPublicKey(ctx, tn, total=5, threshold=3)
sh = [shares[i - 1] for i in (1, 3, 5)]
raw = rg.integers(0, 256, size=(16384, 512), dtype=np.uint8); raw[:, 0] = 0
c = torch.from_numpy(raw).to(dev)
for world in (1, 2, 4, 8):
    cnt = 16384 // world
    o = torch.empty((3 * cnt, 512), dtype=torch.uint8, device=dev)
    out["threshold_2048_replicated"][str(world)] = best(lambda: tk.partial_decrypt_units_raw(sh, cnt, c[:cnt].data_ptr(), 512, 0, 3 * cnt, o.data_ptr(), 512, MEM_DEVICE))
    # the exchange flow: rank r computes units [r * 3 * 16384 / N, (r + 1) * 3 * 16384 / N) of the whole batch (server-major); the slowest rank
    # (one whose range straddles two servers) is the floor
    per = 3 * 16384 // world
    worst = 0.0
    for r in range(world):
        worst = max(worst, best(lambda: tk.partial_decrypt_units_raw(sh, 16384, c.data_ptr(), 512, r * per, (r + 1) * per, o.data_ptr(), 512, MEM_DEVICE)))
    out["threshold_2048"][str(world)] = worst
out["_comment"] = ("What ONE rank of a strong-scaled bench entry takes on its share (16384 / N statements or ciphertexts), measured on one GPU by "
                   "tools/rank_floor.py (best of %d calls).  bench.py --gpus N prints the figure for its N beside the entry (predicted_rank_floor_ms): "
                   "these entries are latency-bound by construction -- a ladder's length, not its width, is the run time of a small shard -- so an "
                   "N-GPU run cannot beat N x (job / floor)." % REPS)
for k_, u in (("nested_randomize_2048", "ms per rank for 16384 / N ciphertexts"), ("ddleq_prove_2048", "ms per rank for 16384 / N instances"),
              ("ddleq_verify_2048", "ms per rank for 16384 / N instances"),
              ("threshold_2048", "ms of partial decryption on the slowest rank: its range of the 3 x 16384 (server, ciphertext) units; the all-gather of the partials and the combine of 16384 / N ciphertexts come on top"),
              ("threshold_2048_replicated", "ms of partial decryption per rank: a slice of 16384 / N ciphertexts under all three shares (no exchange); the combine comes on top (4 - 5 ms at N = 1)")):
    out[k_]["unit"] = u
print(json.dumps(out, indent=1))
