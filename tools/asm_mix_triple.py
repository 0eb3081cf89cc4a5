#!/usr/bin/env python3
"""Instruction mix of the multi-lane digit kernels, counted from the generator's own output: issue slots per squaring and per
product (loop bodies times their trip counts).  At one wave per SIMD every instruction -- scalar, LDS and waits included --
takes an issue slot of its own.
  asm_mix_triple.py [H]        the three-digit kernel vm_asm_<H>_48 (level-two Encrypt / NestedRandomize / DDLEQ; default 74)
  asm_mix_triple.py 37 64      the four-lane pair kernel vm_asm_37_64 (PartialDecrypt / the mod-n^2 stages at 16 384 numbers)
  asm_mix_triple.py 74 32      the two-lane pair kernel vm_asm_74_32 (Encrypt-2048)
  asm_mix_triple.py 19 96      the eight-lane pair kernel; 55 112 / 37 112: the three-digit kernel with two lanes per digit"""
import os, sys, re, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "paillier_amd", "csrc"))
import gen_vm_asm

H = int(sys.argv[1]) if len(sys.argv) > 1 else 74
TAG = int(sys.argv[2]) if len(sys.argv) > 2 else 48
GEN = gen_vm_asm.make_gen(H, TAG)
lines = GEN.generate().splitlines()
ROWS = {48: H, 64: 2 * H, 32: H, 96: 4 * H, 112: 2 * H}[TAG]                      # rows of a pass = limbs of a digit


def classify(seg):
    c = collections.Counter()
    for ln in seg:
        ln = ln.strip()
        if not ln or ln.endswith(":") or ln.startswith(".") or ln.startswith("//") or ln.startswith(";"):
            continue
        op = ln.split()[0]
        if op in ("v_mad_u64_u32",):
            c["mad"] += 1
        elif op.startswith("v_"):
            c["valu_other"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op in ("s_waitcnt", "s_nop"):
            c["wait"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def find(label):
    return next(i for i, l in enumerate(lines) if l.strip() == label + ":")


def region(start, end_label_branch):
    """dynamic counts of the code from `start` to the first `s_branch L_next` after it; a loop L_q* runs as often as its counter says
    (s_mov_b32 s19, <first> before the label; s_add_u32 s19, s19, <step> and s_cmp_lt_u32 s19, <limit> inside)"""
    i = find(start)
    j = next(k for k in range(i, len(lines)) if lines[k].strip() == "s_branch L_next")
    tot = collections.Counter()
    k = i
    first = 0
    while k < j:
        m0 = re.match(r"^s_mov_b32 s19, (\d+)$", lines[k].strip())
        if m0:
            first = int(m0.group(1))
        m = re.match(r"^(L_q\w+):", lines[k].strip())
        if m:
            lbl = m.group(1)
            e = next(x for x in range(k, j) if lines[x].strip() == f"s_cbranch_scc1 {lbl}")
            body_lines = [l.strip() for l in lines[k:e + 1]]
            step = next(int(re.match(r"^s_add_u32 s19, s19, (\d+)$", l).group(1)) for l in body_lines if re.match(r"^s_add_u32 s19, s19, \d+$", l))
            limit = next(int(re.match(r"^s_cmp_lt_u32 s19, (\d+)$", l).group(1)) for l in body_lines if re.match(r"^s_cmp_lt_u32 s19, \d+$", l))
            trips = -(-(limit - first) // step)
            body = classify(lines[k:e + 1])
            for key, v in body.items():
                tot[key] += v * trips
            k = e + 1
        else:
            for key, v in classify([lines[k]]).items():
                tot[key] += v
            k += 1
    return tot


out = {}
# multiplies the algorithm needs, per lane of a number: three-digit kernel 8 H^2 / 12 H^2 over 4 lanes; pair kernels (digit of
# D limbs) 4 D^2 / 6 D^2 over 4 lanes (GenQ4: one-pass product) or 4 D^2 / 6 D^2 needed of 4 D^2 / 8 D^2 executed over 2 lanes (GenQ)
D = ROWS
USEFUL = {48: (8 * H * H / 4, 12 * H * H / 4), 64: (4 * D * D / 4, 6 * D * D / 4), 32: (4 * D * D / 2, 6 * D * D / 2),
          96: (4 * D * D / 8, 6 * D * D / 8), 112: (8 * D * D / 8, 12 * D * D / 8)}[TAG]
for name, lbl, useful in (("squaring", "L_montsq", USEFUL[0]), ("product", "L_montmul", USEFUL[1])):
    t = region(lbl, None)
    slots = sum(t.values())
    out[name] = dict(t, issue_slots=slots, valu=t["mad"] + t["valu_other"], counted_mads_per_lane=useful,
                     counted_share_of_issue_slots=round(useful / slots, 4), mad_share_of_valu=round(t["mad"] / (t["mad"] + t["valu_other"]), 4))
print(json.dumps({"kernel": f"vm_asm_{H}_{TAG}", "vgprs": GEN.n_vgpr, "lds_bytes": GEN.lds_bytes, **out}, indent=1))
