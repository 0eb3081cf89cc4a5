#!/usr/bin/env python3
"""DDLEQ prover alone (16384 instances, 2048-bit), for profiling: prints the time of each call.
   prove_only.py [instances] [secpar]     secpar > 1: instances / secpar statements through pgpu_ddleq_prove_secpar
   PGPU_PROFILE_DUMP=1 lists every profiled VM launch of a call."""
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
for flag in ("early", "side", "nm4", "handover", "struct", "exclusive", "background", "exclusive_short", "spread", "lift", "late", "base_early", "prime_lanes"):                            # PROVE_EARLY=0 etc.: A/B runs of the prover's switches
    if os.environ.get("PROVE_" + flag.upper()) is not None:
        ctx.set_flag(flag, int(os.environ["PROVE_" + flag.upper()]))
SP = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BI = int(sys.argv[1]) if len(sys.argv) > 1 else 16384          # instances
B = BI // SP                                                    # statements
BI = B * SP
rg = np.random.default_rng(int(os.environ.get("PROVE_SEED", "5")))
def below(mod, nb, cnt=None):
    raw = rg.integers(0, 256, size=(cnt or B, nb), dtype=np.uint8); raw[:, 0] %= np.uint8(max(1, min(255, mod >> (8 * (nb - 1))))); return raw
def unit(cnt=None):
    a = below(n, 256, cnt); a[:, -1] |= 1; return a
tb = lambda a: torch.from_numpy(a).to(dev)
msg, r1, r2, a_, b_, x_, y_ = below(n, 256), unit(), unit(), unit(), unit(), unit(BI), unit(BI)
inner = torch.zeros((B, 512), dtype=torch.uint8, device=dev); ct1 = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
pk.encrypt_with_r_raw(B, tb(msg).data_ptr(), 256, tb(r1).data_ptr(), 256, inner.data_ptr(), 512, MEM_DEVICE)
pk.encrypt_with_r_raw(B, inner.data_ptr(), 512, tb(r2).data_ptr(), 256, ct1.data_ptr(), 768, MEM_DEVICE, level=1)
m2, m3 = pa.Modulus(ctx, n * n), pa.Modulus(ctx, n ** 3)
da, db, dx, dy = tb(a_), tb(b_), tb(x_), tb(y_)
an = torch.zeros((B, 512), dtype=torch.uint8, device=dev); t3 = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
bn2 = torch.zeros((B, 768), dtype=torch.uint8, device=dev); ct2 = torch.zeros((B, 768), dtype=torch.uint8, device=dev)
m2.exp_raw(B, da.data_ptr(), 256, n.to_bytes(256, "big"), 256, 0, an.data_ptr(), 512, MEM_DEVICE)
m3.exp_raw(B, ct1.data_ptr(), 768, an.data_ptr(), 512, 512, t3.data_ptr(), 768, MEM_DEVICE)
m3.exp_raw(B, db.data_ptr(), 256, (n * n).to_bytes(512, "big"), 512, 0, bn2.data_ptr(), 768, MEM_DEVICE)
m3.mul_raw(B, t3.data_ptr(), 768, bn2.data_ptr(), 768, ct2.data_ptr(), 768, MEM_DEVICE)
al = torch.zeros((BI, 768), dtype=torch.uint8, device=dev); pe = torch.zeros((BI, 512), dtype=torch.uint8, device=dev); pf = torch.zeros((BI, 768), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
print("PROVE_BEGIN", flush=True)
for _ in range(int(os.environ.get("PROVE_REPS", "2"))):
    t = time.perf_counter()
    if SP == 1:
        sk.ddleq_prove_raw(B, ct1.data_ptr(), ct2.data_ptr(), da.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr(), al.data_ptr(), pe.data_ptr(), pf.data_ptr(), MEM_DEVICE)
    else:
        sk.ddleq_prove_secpar_raw(B, SP, ct1.data_ptr(), ct2.data_ptr(), da.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr(), al.data_ptr(), pe.data_ptr(), pf.data_ptr(), MEM_DEVICE)
    print("prove", B, "x", SP, (time.perf_counter() - t) * 1e3, "ms", ctx.last_profile(), "all VM launches of the call:", ctx.last_vm_launches(), flush=True)
if os.environ.get("PROVE_VERIFY"):
    # every instance against its statement (device-resident pgpu_ddleq_verify; statement rows repeated per instance)
    rep = lambda tns: tns.repeat_interleave(SP, dim=0).contiguous() if SP > 1 else tns
    c1r, c2r = rep(ct1), rep(ct2)
    ok = np.zeros(BI, dtype=np.int32)
    t = time.perf_counter()
    pk.ddleq_verify_raw(BI, c1r.data_ptr(), c2r.data_ptr(), dx.data_ptr(), dy.data_ptr(), al.data_ptr(), pe.data_ptr(), pf.data_ptr(), ok, MEM_DEVICE)
    print("verify", BI, (time.perf_counter() - t) * 1e3, "ms; accepted", int(ok.sum()), "of", BI, ctx.last_profile(), flush=True)
    assert int(ok.sum()) == BI
