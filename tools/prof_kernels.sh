#!/bin/bash
# per-kernel time breakdown of one bench.py run (rocprofv3 kernel trace); run on the GPU box:  bash tools/prof_kernels.sh <tag>
set -e
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o b -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$tag.log 2>&1
f=$(find gpurun_out/$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f'{r["Name"][:46]:46s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:9.3f} avg_us {float(r["AverageNs"])/1e3:10.1f} {r["Percentage"]:>6s}%')
PY
