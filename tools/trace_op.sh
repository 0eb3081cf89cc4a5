#!/bin/bash
# kernel trace of one op of tools/ops_probe.py; run on the GPU box:  bash tools/trace_op.sh <op> [batch] [window_ms]
set -e
op=$1; b=${2:-}; win=${3:-10}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o/kt_$op -o t -- python3 tools/ops_probe.py $op $b 2 > $o/kt_$op.log 2>&1
f=$(find $o/kt_$op -name '*kernel_trace.csv' | head -1)
cp "$f" $o/trace_$op.csv; rm -rf $o/kt_$op
python3 tools/trace_gaps.py $o/trace_$op.csv $win > $o/gaps_$op.txt
grep "^$op" $o/kt_$op.log | tail -2
