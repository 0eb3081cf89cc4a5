"""Diff the assembly VM kernel against the hipcc one, opcode by opcode (run on the GPU box)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import paillier_amd as pa
from model28 import to_limbs, from_limbs, MASK
END, LOAD, STORE, LOADC, SQR, MUL, MULC, MULV, ADD = 0, 1, 2, 3, 4, 5, 6, 7, 8
shapes = {1024: (37, 1), 1536: (55, 1), 2048: (74, 1), 3072: (55, 2), 4096: (74, 2), 6144: (55, 4)}  # WT = wl*k; small nb -> run_vm re-slices 74->37x2, 148->37x4
ctx = pa.Context(0)
bits_list = [int(x) for x in sys.argv[1:]] or [1024]
for bits in bits_list:
    wl, k = shapes[bits]; wt = wl * k
    rng = random.Random(bits)
    n = rng.getrandbits(bits) | 1 | (1 << (bits - 1))
    mod = pa.Modulus(ctx, n)
    nb, nslots = 256, 4
    mem = np.zeros((nslots, wt, nb), dtype=np.uint32)
    for g in range(nb):
        for s in range(2):
            mem[s, :, g] = to_limbs(rng.randrange(n), wt)
    progs = {
        "load-store": [LOAD, 0, STORE, 2, END, 0],
        "loadc-store": [LOADC, 0, STORE, 2, LOADC, 2, STORE, 3, END, 0],
        "add": [LOAD, 0, ADD, 1, STORE, 2, END, 0],
        "mulc": [LOAD, 0, MULC, 0, STORE, 2, END, 0],
        "mul": [LOAD, 0, MUL, 1, STORE, 2, END, 0],
        "sqr": [LOAD, 0, SQR, 0, STORE, 2, END, 0],
        "sqr2": [LOAD, 0, SQR, 0, SQR, 0, STORE, 2, MUL, 1, STORE, 3, END, 0],
    }
    for name, pr in progs.items():
        a = mod.vm_debug_run(pr, mem, nslots, nb, True)
        c = mod.vm_debug_run(pr, mem, nslots, nb, False)
        same = (a == c).all()
        if not same:
            # different lane slicings may leave different (equally valid) lazy limb forms: compare the integers
            bad = [(s_, g_) for s_ in range(nslots) for g_ in range(nb)
                   if from_limbs(a[s_, :, g_]) != from_limbs(c[s_, :, g_])]
            if not bad:
                print(f"{bits} {name}: OK (same integers, different limb form)")
                continue
            print(f"   {len(bad)} numbers differ as integers; first (slot, number): {bad[:4]}")
        print(f"{bits} {name}: {'OK' if same else 'DIFF'}")
        if not same:
            idx = np.argwhere(a != c)
            print("   first diffs (slot, limb, number):", idx[:6].tolist(), "count", len(idx), "of", a.size)
            s_, l_, g_ = idx[0]
            print("   asm:", [hex(int(v)) for v in a[s_, max(0,l_-1):l_+3, g_]], " cc:", [hex(int(v)) for v in c[s_, max(0,l_-1):l_+3, g_]])
            lanes = sorted(set(int(i[2]) for i in idx)); limbs = sorted(set(int(i[1]) for i in idx))
            print("   lanes affected:", len(lanes), lanes[:10], " limbs affected:", len(limbs), limbs[:12])
            va = from_limbs(a[s_, :, g_]); vc = from_limbs(c[s_, :, g_])
            print("   value equal mod n:", (va - vc) % n == 0, " asm<2n:", va < 2*n)
