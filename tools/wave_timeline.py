#!/usr/bin/env python3
"""Per-wave timeline of the headline kernel (vm_asm_37_16) on a DEBUG build of the library (generator run with
PGPU_GEN_TIMING=1: every wave writes the real-time counter at its start and at END, HW_ID and XCC_ID into slot 0).
Runs a Decrypt-shaped ladder (32-entry table, 1024 squarings, a product every 7th) on `nb` numbers in one segment through
pgpu_pair_debug_run and prints when waves start and finish, by XCD.   usage: wave_timeline.py [nb=131072] [squarings=1024] [priority levels=4]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import paillier_amd as pa
END, LOAD, STORE, SQR, MUL = 0, 1, 2, 4, 5
K = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
p = int(K["p"], 16)
H = 37
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
nsq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = pa.Context(0)
nslots = 37
rng = np.random.default_rng(3)
mem = np.zeros((nslots, 2 * H, nb), dtype=np.uint32)
mem[0] = rng.integers(0, 1 << 27, size=(2 * H, nb), dtype=np.uint32)
prog = [LOAD, 0, STORE, 4, SQR, 0, STORE, 2, LOAD, 4]
for k in range(1, 32):
    prog += [MUL, 2, STORE, 4 + k]
prog += [LOAD, 9]
for i in range(nsq):
    prog += [SQR, 0]
    if i % 7 == 6:
        prog += [MUL, 4 + (i * 5) % 32]
prog += [STORE, 3, END, 0]
levels = int(sys.argv[3]) if len(sys.argv) > 3 else 4       # priority steps over the program (1: none -- oldest-first only)
if levels > 1:
    nops = len(prog) // 2
    cuts = [0.80, 0.96, 0.992] if levels == 5 else [(k + 1) / levels for k in range(levels - 1)]     # 5: the geometric stretches of Prog::end
    for i in range(nops):
        prog[2 * i] |= (3 - min(3, sum(i / nops >= c for c in cuts))) << 30
for rep in range(2):
    out, _, _ = ctx.pair_debug_run(p, prog, mem, nslots, nb)
nw = nb // 64
rec = out[0, 0, :nw * 8].reshape(nw, 8).astype(np.uint64)
start = rec[:, 0] | (rec[:, 1] << np.uint64(32))
end = rec[:, 2] | (rec[:, 3] << np.uint64(32))
hw, xcc = rec[:, 4].astype(np.int64), rec[:, 5].astype(np.int64) & 0xF
t0 = start.min()
s_us = (start - t0).astype(np.float64) / 100.0
e_us = (end - t0).astype(np.float64) / 100.0
dur = e_us - s_us
print(f"waves {nw}: kernel span {e_us.max() / 1e3:.3f} ms; start min/median/max {s_us.min():.1f}/{np.median(s_us):.1f}/{s_us.max():.1f} us; "
      f"end min/median/max {e_us.min() / 1e3:.3f}/{np.median(e_us) / 1e3:.3f}/{e_us.max() / 1e3:.3f} ms; "
      f"life min/median/max {dur.min() / 1e3:.3f}/{np.median(dur) / 1e3:.3f}/{dur.max() / 1e3:.3f} ms")
late = s_us > 100
print(f"waves starting later than 100 us: {int(late.sum())} (their median start {np.median(s_us[late]) / 1e3 if late.any() else 0:.3f} ms)")
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  XCC {x}: waves {int(m.sum()):5d}  late starters {int((late & m).sum()):4d}  end median {np.median(e_us[m]) / 1e3:.3f} max {e_us[m].max() / 1e3:.3f} ms  "
          f"life median {np.median(dur[m]) / 1e3:.3f} ms  CU slots used {len(set(zip(se[m].tolist(), cu[m].tolist())))}")
# waves per (xcc, se, cu, simd) among the early starters
from collections import Counter
c = Counter(zip(xcc[~late].tolist(), se[~late].tolist(), cu[~late].tolist(), simd[~late].tolist()))
print("early starters per SIMD: histogram of counts", sorted(Counter(c.values()).items()), " SIMDs seen", len(c))
hist, edges = np.histogram(e_us / 1e3, bins=12)
print("end-time histogram (ms):", [(round(float(a), 2), int(b)) for a, b in zip(edges[:-1], hist)])
# the two waves of each SIMD: when does the first one end, when the second?
from collections import defaultdict
by = defaultdict(list)
for i in range(nw):
    by[(int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i]))].append(e_us[i] / 1e3)
firsts = np.array([min(v) for v in by.values() if len(v) == 2])
seconds = np.array([max(v) for v in by.values() if len(v) == 2])
if len(firsts):
    print(f"SIMDs with two waves: {len(firsts)}; first wave ends min/median/max {firsts.min():.2f}/{np.median(firsts):.2f}/{firsts.max():.2f} ms; "
          f"second wave ends {seconds.min():.2f}/{np.median(seconds):.2f}/{seconds.max():.2f} ms")
# per CU: end of its last wave
bycu = defaultdict(list)
for i in range(nw):
    bycu[(int(xcc[i]), int(se[i]), int(cu[i]))].append(e_us[i] / 1e3)
cu_end = np.array([max(v) for v in bycu.values()])
print(f"CUs {len(cu_end)}: last wave ends min/median/max {cu_end.min():.2f}/{np.median(cu_end):.2f}/{cu_end.max():.2f} ms; waves per CU {sorted(Counter(len(v) for v in bycu.values()).items())}")
