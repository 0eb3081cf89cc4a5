#!/usr/bin/env python3
"""Print the headline and the extra_configs of a bench.py log (one JSON line) as a table.

  bench_summary.py <log>                 the table (+ the plan guard's verdict)
  bench_summary.py --plan-table <log>    profiles/plan_table.json for this run's executed multiply-adds per unit -- regenerate it
                                         ONLY together with a deliberate change of the plan (paillier_amd/csrc/plan.hpp)
"""
import json
import sys


def last_line(path):
    j = None
    for l in open(path):
        if l.startswith('{'):
            j = json.loads(l)
    if j is None:
        sys.exit(f"{path}: no JSON line")
    return j


if len(sys.argv) >= 3 and sys.argv[1] == "--plan-table":
    j = last_line(sys.argv[2])
    per = {"headline_decrypt_2048": j["roofline"]["alg_mad28_per_decrypt"]}
    for e in j.get("extra_configs", []):
        per[e["config"]] = e.get("executed_mad28_per_unit") or e["executed_mad28"] / round(e["value"] * e["ms_per_batch"] / 1e3)
    import os
    keep = {}
    try:      # the yardstick of the literal algorithm is measured with "struct" off, not by this run: carried over
        keep = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "plan_table.json"))).get("literal_per_unit", {})
    except OSError:
        pass
    print(json.dumps({
        "_comment": "Executed 28-bit multiply-adds per unit of every bench config on ONE GPU at the default shapes (bench.py asserts "
                    "its own counts against these, +-1 %, and prints plan_changed otherwise): a planning predicate that silently "
                    "moves a BASELINE shape onto another ladder changes these numbers before it changes a timing.  Regenerate ONLY "
                    "together with a deliberate change of the plan: python tools/bench_summary.py --plan-table <bench line> > "
                    "profiles/plan_table.json",
        "tolerance": 0.01, "source": sys.argv[2], "per_unit": per, "literal_per_unit": keep}, indent=1))
    sys.exit(0)

j = last_line(sys.argv[1])
print(f"headline {j['value']:.0f} {j['unit']}  frac {j['roofline']['frac']:.4f}  kernel_ms {j['roofline']['kernel_ms_per_launch']:.2f}")
for e in j.get('extra_configs', []):
    fr, fc, cpu = e.get('frac'), e.get('frac_of_call_time'), e.get('cpu_per_s')
    print(f"{e['config']:26s} {e['value']:10.0f}  ms {e['ms_per_batch']:8.2f}  kernel_ms {e['kernel_ms_per_batch']:8.2f}  "
          f"frac {fr if fr is None else round(fr, 4)}  of_call {fc if fc is None else round(fc, 4)}"
          + (f"  cpu {cpu:.1f}/s" if cpu else ""))
if "plan_changed" in j:
    print("plan_changed", j["plan_changed"], j.get("plan_deviations") or "")
