#!/bin/bash
# PMC passes for the dominant kernel of bench.py (separate rocprofv3 --pmc runs, no tracing domains); run on the GPU box:
#   bash tools/prof_pmc.sh <tag> <kernel-name>
set -e
tag=${1:-pmc}; kern=${2:-vm_asm_37_16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
# the TCC byte counters and GRBM each need a pass of their own ("exceeds the capabilities of the hardware" together)
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_$i.log 2>&1
done
python3 - "$tag" "$kern" <<'PY'
import csv, glob, sys, collections
tag, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(kern):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
with open(f"gpurun_out/{tag}_summary.txt", "w") as o:
    for c in sorted(acc):
        line = f"{c}: {sum(acc[c]) / len(acc[c]):.6g}  (mean of {len(acc[c])} dispatches)"
        print(line); o.write(line + "\n")
PY
