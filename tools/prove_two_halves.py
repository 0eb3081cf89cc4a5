#!/usr/bin/env python3
"""Experiment: the 16384-instance prove job as TWO concurrent calls of 8192 on two contexts (own streams, one host thread each) against one
call of 16384 -- does a second call in flight fill the issue slots the first one leaves (its one-wave-per-SIMD response ladder, the chains)?
   prove_two_halves.py [instances] [parts]"""
import json, os, sys, time, threading
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
BI = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
PARTS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
OFFSET_MS = float(os.environ.get("OFFSET_MS", "0"))
rg = np.random.default_rng(5)
def below(mod, nb, cnt):
    raw = rg.integers(0, 256, size=(cnt, nb), dtype=np.uint8); raw[:, 0] %= np.uint8(max(1, min(255, mod >> (8 * (nb - 1))))); return raw
def unit(cnt):
    a = below(n, 256, cnt); a[:, -1] |= 1; return a
tb = lambda a: torch.from_numpy(a).to(dev)
class Job:
    def __init__(self, B):
        self.B = B
        self.ctx = pa.Context(0)                      # own stream
        self.pk = pa.PublicKey(self.ctx, n); self.sk = pa.SecretKey(self.ctx, self.pk, (p - 1) * (q - 1))
        msg, r1, r2, a_, b_, x_, y_ = below(n, 256, B), unit(B), unit(B), unit(B), unit(B), unit(B), unit(B)
        inner = torch.zeros((B, 512), dtype=torch.uint8, device=dev); self.ct1 = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
        self.pk.encrypt_with_r_raw(B, tb(msg).data_ptr(), 256, tb(r1).data_ptr(), 256, inner.data_ptr(), 512, MEM_DEVICE)
        self.pk.encrypt_with_r_raw(B, inner.data_ptr(), 512, tb(r2).data_ptr(), 256, self.ct1.data_ptr(), 768, MEM_DEVICE, level=1)
        self.da, self.db, self.dx, self.dy = tb(a_), tb(b_), tb(x_), tb(y_)
        self.ct2 = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
        self.pk.nested_randomize_with_ab_raw(B, self.ct1.data_ptr(), self.da.data_ptr(), self.db.data_ptr(), self.ct2.data_ptr(), MEM_DEVICE)
        self.al = torch.zeros((B, 768), dtype=torch.uint8, device=dev); self.pe = torch.zeros((B, 512), dtype=torch.uint8, device=dev); self.pf = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
    def prove(self):
        self.sk.ddleq_prove_raw(self.B, self.ct1.data_ptr(), self.ct2.data_ptr(), self.da.data_ptr(), self.db.data_ptr(), self.dx.data_ptr(), self.dy.data_ptr(),
                                self.al.data_ptr(), self.pe.data_ptr(), self.pf.data_ptr(), MEM_DEVICE)
    def verify(self):
        ok = np.zeros(self.B, dtype=np.int32)
        self.pk.ddleq_verify_raw(self.B, self.ct1.data_ptr(), self.ct2.data_ptr(), self.dx.data_ptr(), self.dy.data_ptr(), self.al.data_ptr(), self.pe.data_ptr(), self.pf.data_ptr(), ok, MEM_DEVICE)
        return int(ok.sum())
whole = Job(BI)
parts = [Job(BI // PARTS) for _ in range(PARTS)]
torch.cuda.synchronize()
def run_parts():
    def work(j, k):
        if OFFSET_MS and k: time.sleep(OFFSET_MS * k / 1e3)
        j.prove()
    th = [threading.Thread(target=work, args=(j, k)) for k, j in enumerate(parts)]
    for t in th: t.start()
    for t in th: t.join()
for rep in range(4):
    t = time.perf_counter(); whole.prove(); t1 = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); run_parts(); t2 = (time.perf_counter() - t) * 1e3
    print(f"one call of {BI}: {t1:.1f} ms | {PARTS} concurrent calls of {BI // PARTS}: {t2:.1f} ms", flush=True)
print("verified:", whole.verify(), [j.verify() for j in parts])
