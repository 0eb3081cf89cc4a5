#!/bin/bash
# kernel timeline of the last headline step (Decrypt-2048 x 65536): what runs besides the ladder kernel, and the gaps.
#   bash tools/prof_step.sh <tag>      (on the GPU box)
set -e
tag=${1:-step}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -o t -- python3 bench.py --no-extra --no-traffic --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/$tag.log 2>&1
f=$(find gpurun_out/$tag -name '*kernel_trace.csv' | head -1)
python3 tools/trace_gaps.py "$f" ${2:-25.6} > gpurun_out/${tag}_gaps.txt
cp "$f" gpurun_out/${tag}_trace.csv; rm -rf gpurun_out/$tag
cat gpurun_out/${tag}_gaps.txt
