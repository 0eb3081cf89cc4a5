import json, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import paillier_amd as pa
from model28 import to_limbs, from_limbs
END, LOAD, STORE, SQR, MUL = 0, 1, 2, 4, 5
LB = 28
k = json.load(open(os.path.join(ROOT, "tests/golden/keys.json")))["paillier"]["2048"]
n = int(k["p"], 16) * int(k["q"], 16)
ctx = pa.Context(0)
rng = random.Random(3)
nb, nslots, H = 256, 4, 74
mem = np.zeros((nslots, 2 * H, nb), dtype=np.uint32)
vals = {}
for s in (0, 1):
    for g in range(nb):
        a = (rng.randrange(n), rng.randrange(n))
        vals[s, g] = a
        mem[s, :H, g] = to_limbs(a[0], H); mem[s, H:, g] = to_limbs(a[1], H)
prog = [LOAD, 0, MUL, 1, STORE, 2, END, 0]
out, consts, h = ctx.pair_debug_run(n, prog, mem, nslots, nb, lanes=2)
cadj = from_limbs(consts[H:]); R = 1 << (LB * H); nneg = (-pow(n, -1, R)) % R
def mont(u):
    m = (u * nneg) % R
    return (u + m * n) // R, m
for g in (0, 1, 2, 63, 64, 255):
    a, b = vals[0, g], vals[1, g]
    r2, _ = mont(a[0] * b[1]); t, m = mont(a[0] * b[0]); r1, _ = mont(a[1] * b[0] + cadj - m)
    got = (from_limbs(out[2, :H, g]), from_limbs(out[2, H:, g]))
    d = got[1] - (r1 + r2)
    cands = {"0": 0, "-r2": -r2, "+r2": r2, "-r1": -r1, "r2'(a1*b1)": mont(a[1] * b[1])[0] - r2, "garbage lane1 pass1 = a1*b1": None}
    print(g, "t ok:", got[0] == t, "c1 diff:", [kname for kname, v in cands.items() if v is not None and v == d], "diff bits", d.bit_length() if d else 0,
          "got1==r1", got[1] == r1, "got1==r1+mont(a1*b1)", got[1] == r1 + mont(a[1] * b[1])[0])
