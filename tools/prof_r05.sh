#!/bin/bash
# Round-5 profiles (run on the GPU box): rocprofv3 kernel trace + stats of the whole bench command, and PMC passes for the
# dominant kernel of every config (separate --pmc runs, no tracing domains combined with them).
#   bash tools/prof_r05.sh [kernel-trace|pmc]
set -e
what=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r05
mkdir -p $out
if [ "$what" = all ] || [ "$what" = kernel-trace ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 5 --warmup 2 --no-traffic --no-cpu-baseline > $out/kt.log 2>&1
  f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
  cp "$f" $out/bench_kernel_stats.csv
  head -25 $out/bench_kernel_stats.csv
fi
if [ "$what" = all ] || [ "$what" = headline ]; then
  # the headline alone: in the full run the same kernel also serves the DDLEQ prover (a^n | x^n modulo p^2, q^2), which
  # would blur its average duration
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kh -o kh -- python3 bench.py --steps 20 --warmup 5 --no-extra --no-traffic --no-cpu-baseline > $out/kh.log 2>&1
  f=$(find $out/kh -name '*kernel_stats.csv' | head -1)
  cp "$f" $out/headline_kernel_stats.csv
  head -4 $out/headline_kernel_stats.csv
  tail -1 $out/kh.log | cut -c1-400
fi
if [ "$what" = all ] || [ "$what" = pmc ]; then
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    echo "pmc pass $i: $set"
    timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-traffic --no-cpu-baseline --extra-steps 1 > $out/pmc_$i.log 2>&1
  done
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
kernels = ["vm_asm_37_16", "vm_asm_74_32", "vm_asm_55_16", "vm_asm_74_48", "vm_asm_37_48", "vm_asm_37_64", "vm_asm_37_1"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        for k in kernels:
            if r["Kernel_Name"] in (k, k + ".kd"):          # (exact: vm_asm_37_1 is a prefix of vm_asm_37_16)
                # one entry per launch shape: the same kernel runs 65536- and 131072-ciphertext batches in bench.py
                kk = f'{k} grid {r["Grid_Size"]}' if "Grid_Size" in r else k
                per[(kk, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
with open(f"{out}/bench_pmc_summary.txt", "w") as o:
    for k in sorted(acc):
        if not acc[k]:
            continue
        o.write(f"[{k}]  (per dispatch; median over the dispatches of the run; sums over all SEs / XCDs)\n")
        for c in sorted(acc[k]):
            v = sorted(acc[k][c])
            o.write(f"  {c}: median {v[len(v)//2]:.6g}  min {v[0]:.6g}  max {v[-1]:.6g}  ({len(v)} dispatches)\n")
print(open(f"{out}/bench_pmc_summary.txt").read())
PY
fi
