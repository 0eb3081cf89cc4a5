#!/usr/bin/env python3
"""Instruction mix of the pair kernel vm_asm_37_16 (the headline Decrypt-2048 kernel), counted from the generator's own
output: per squaring and per product, dynamic counts (loop bodies times their trip counts)."""
import os, sys, re, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "paillier_amd", "csrc"))
import gen_vm_asm

H = 37
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
        elif op == "s_waitcnt" or op == "s_nop":
            c["wait"] += 1
        elif op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64"):
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def region(text, start, loop, trips):
    body = text.split(start)[1].split("s_branch L_next")[0]
    pre, rest = body.split(loop + ":")
    lp, post = rest.split("s_cbranch_scc1 " + loop)
    tot = {}
    for part, mult in ((pre, 1), (lp + "s_cbranch_scc1 x", trips), (post, 1)):
        for k, v in classify(part.splitlines()).items():
            tot[k] = tot.get(k, 0) + v * mult
    return tot


out = {}
for name, start, loop in (("squaring", "L_montsq:", "L_p2s"), ("product", "L_montmul:", "L_p2m")):
    t = region(text, start, loop, (H - 1) // 2)
    valu = t["mad"] + t["valu4"] + t["valu2"]
    t["valu_total"] = valu
    t["mad_share_of_valu"] = round(t["mad"] / valu, 4)
    # issue cost in quarter-rate slots: full-rate instructions cost half a slot
    t["mad_share_of_issue_slots"] = round(t["mad"] / (t["mad"] + t["valu4"] + 0.5 * t["valu2"]), 4)
    out[name] = t
# a Decrypt-2048 half: 1025 squarings + 176 products (sliding windows over p - 1)
sq, mu = out["squaring"], out["product"]
mix = {k: 1025 * sq[k] + 176 * mu[k] for k in ("mad", "valu4", "valu2", "salu", "lds", "wait", "branch")}
mix["mad_share_of_valu"] = round(mix["mad"] / (mix["mad"] + mix["valu4"] + mix["valu2"]), 4)
mix["mad_share_of_issue_slots"] = round(mix["mad"] / (mix["mad"] + mix["valu4"] + 0.5 * mix["valu2"]), 4)
out["decrypt_2048_half_ladder"] = mix
print(json.dumps(out, indent=1))
