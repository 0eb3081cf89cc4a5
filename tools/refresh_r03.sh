#!/bin/bash
# Round-3 side measurements in one go (run on the GPU box); outputs under gpurun_out/refresh/
mkdir -p gpurun_out/refresh; o=gpurun_out/refresh
timeout -k 10 200 python3 tools/bench_configs.py l2_3072 2>/dev/null | tail -1 > $o/l2_3072.json &&
timeout -k 10 200 python3 tools/lanes8_probe.py 2>/dev/null | grep -v amdgpu > $o/lanes8.txt &&
timeout -k 10 200 python3 tools/l2_lanes_probe.py 2>/dev/null | grep -v amdgpu > $o/l2_lanes.jsonl &&
timeout -k 10 300 python3 tools/threshold_shard_probe.py 2>/dev/null | grep "^{" > $o/shard.jsonl &&
timeout -k 10 300 python3 tools/small_batch_sweep.py 2>/dev/null | grep '^{' > $o/small.jsonl &&
PGPU_PROFILE_DUMP=1 timeout -k 10 200 python3 tools/prove_only.py 2>&1 | grep -v amdgpu | tail -6 > $o/prove_dump.txt &&
PGPU_PROFILE_DUMP=1 timeout -k 10 200 python3 tools/prove_only.py 61440 40 2>&1 | grep -v amdgpu | tail -7 > $o/prove40_dump.txt
echo rc=$?
