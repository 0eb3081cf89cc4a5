#!/bin/bash
# ordered kernel timeline of a small prove call:  bash tools/prof_small.sh [instances] [window_ms]
set -e
n=${1:-2048}; w=${2:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PROVE_REPS=3 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ps -o t -- python3 tools/prove_only.py $n > gpurun_out/ps.log 2>&1
f=$(find gpurun_out/ps -name '*kernel_trace.csv' | head -1)
python3 tools/trace_list.py "$f" $w 200 > gpurun_out/prove${n}_list.txt
python3 tools/trace_list.py "$f" $w 0 > gpurun_out/prove${n}_list_all.txt
rm -rf gpurun_out/ps
grep '^prove' gpurun_out/ps.log | cut -c1-60; cat gpurun_out/prove${n}_list.txt
