#!/usr/bin/env python3
"""Print the headline and the extra_configs of a bench.py log (one JSON line) as a table."""
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        j = json.loads(l)
        print(f"headline {j['value']:.0f} {j['unit']}  frac {j['roofline']['frac']:.4f}  kernel_ms {j['roofline']['kernel_ms_per_launch']:.2f}")
        for e in j.get('extra_configs', []):
            fr = e.get('frac')
            print(f"{e['config']:24s} {e['value']:10.0f}  ms {e['ms_per_batch']:8.2f}  kernel_ms {e['kernel_ms_per_batch']:8.2f}  frac {fr if fr is None else round(fr, 4)}")
