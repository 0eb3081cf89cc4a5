#!/bin/bash
# same-box A/B of two builds of the library (rule: never compare timings across boxes): usage ab_bench.sh other.so
set -e
cp paillier_amd/libpaillier_hip.so /tmp/lib_new.so
for i in 1 2; do
  cp "$1" paillier_amd/libpaillier_hip.so; python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed 's/^/before /'
  cp /tmp/lib_new.so paillier_amd/libpaillier_hip.so; python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed 's/^/after  /'
done
