#!/usr/bin/env python3
"""Instruction mix of the one-lane pair kernels vm_asm_37_16 (the headline Decrypt-2048 kernel; default) and vm_asm_55_16
(Decrypt-3072: `asm_mix.py 55`), counted from the generator's own output: per squaring and per product, dynamic counts
(loop bodies times their trip counts)."""
import os, sys, re, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "paillier_amd", "csrc"))
import gen_vm_asm

H = int(sys.argv[1]) if len(sys.argv) > 1 else 37
text = gen_vm_asm.make_gen(H, 16).generate()


def classify(lines):
    c = dict(mad=0, valu4=0, valu2=0, salu=0, lds=0, wait=0, branch=0)
    for ln in lines:
        ln = ln.strip()
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        op = ln.split()[0]
        if op == "v_mad_u64_u32":
            c["mad"] += 1
        elif op in ("v_and_b32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_xor_b32", "v_or_b32", "v_not_b32"):
            c["valu2"] += 1        # full-rate class (2 cycles per wave-instruction: profiles/r01_valu_rates.txt)
        elif op.startswith("v_"):
            c["valu4"] += 1        # quarter-rate class: v_mul_lo_u32, v_lshrrev_b64, v_lshl_add_u64, v_mov_b64 ...
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("global_"):
            c["vmem"] = c.get("vmem", 0) + 1
        elif op == "s_waitcnt" or op == "s_nop":
            c["wait"] += 1
        elif op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64"):
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def region(text, start, loops):
    """dynamic instruction counts of one VM operation: straight-line code once, every loop body times its trip count"""
    body = text.split(start)[1].split("s_branch L_next")[0]
    tot = {}

    def add(part, mult):
        for k, v in classify(part.splitlines()).items():
            tot[k] = tot.get(k, 0) + v * mult

    for loop, trips in loops:
        pre, rest = body.split(loop + ":")
        lp, body = rest.split("s_cbranch_scc1 " + loop)
        add(pre, 1)
        add(lp + "s_cbranch_scc1 x", trips)
    add(body, 1)
    return tot


out = {}
g = gen_vm_asm.make_gen(H, 16)
if isinstance(g, gen_vm_asm.GenP2):
    ops = (("squaring", "L_montsq:", [("L_p2s", (H - 1) // 2)]),
           ("product", "L_montmul:", [("L_m1", H // g.RB - 1), ("L_m2", H // g.RB - 1)]))
else:
    ops = (("squaring", "L_montsq:", [("L_p2s", (H - 1) // 2)]), ("product", "L_montmul:", [("L_p2m", (H - 1) // 2)]))
for name, start, loops in ops:
    t = region(text, start, loops)
    valu = t["mad"] + t["valu4"] + t["valu2"]
    t["valu_total"] = valu
    t["mad_share_of_valu"] = round(t["mad"] / valu, 4)
    # issue cost in quarter-rate slots: full-rate instructions cost half a slot
    t["mad_share_of_issue_slots"] = round(t["mad"] / (t["mad"] + t["valu4"] + 0.5 * t["valu2"]), 4)
    out[name] = t
# a Decrypt half: the sliding-window ladder over p - 1 (2048-bit keys: 1025 squarings + 176 products; 3072: 1537 + 250)
NSQ, NMU = (1025, 176) if H == 37 else (1537, 250)
sq, mu = out["squaring"], out["product"]
mix = {k: NSQ * sq.get(k, 0) + NMU * mu.get(k, 0) for k in ("mad", "valu4", "valu2", "salu", "lds", "vmem", "wait", "branch")}
mix["mad_share_of_valu"] = round(mix["mad"] / (mix["mad"] + mix["valu4"] + mix["valu2"]), 4)
mix["mad_share_of_issue_slots"] = round(mix["mad"] / (mix["mad"] + mix["valu4"] + 0.5 * mix["valu2"]), 4)
out["decrypt_2048_half_ladder" if H == 37 else "decrypt_3072_half_ladder"] = mix
print(json.dumps(out, indent=1))
