#!/bin/bash
# rocprofv3 kernel-trace of the side configs (tools/bench_configs.py); run on the GPU box:  bash tools/prof_configs.sh <tag> config...
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o c -- python3 tools/bench_configs.py "$@" > gpurun_out/$tag.log 2>&1
f=$(find gpurun_out/$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(f'{r["Name"][:46]:46s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:9.3f} avg_us {float(r["AverageNs"])/1e3:10.1f} {r["Percentage"]:>6s}%')
PY
