#!/bin/bash
# kernel trace of the DDLEQ prover alone; run on the GPU box:  bash tools/prof_prove.sh <tag>
set -e
tag=${1:-prove}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -o t -- python3 tools/prove_only.py $PROVE_ARGS > gpurun_out/$tag.log 2>&1
f=$(find gpurun_out/$tag -name '*kernel_trace.csv' | head -1)
python3 tools/trace_gaps.py "$f" ${2:-300} > gpurun_out/${tag}_gaps.txt
cp "$f" gpurun_out/${tag}_trace.csv; rm -rf gpurun_out/$tag
cat gpurun_out/${tag}_gaps.txt
tail -3 gpurun_out/$tag.log
