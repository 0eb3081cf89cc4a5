#!/bin/bash
# same-box A/B of two builds of the library on the side configs: usage ab_configs.sh other.so config...
set -e
other=$1; shift
cp paillier_amd/libpaillier_hip.so /tmp/lib_new.so
for i in 1 2; do
  cp "$other" paillier_amd/libpaillier_hip.so; python tools/bench_configs.py "$@" 2>/dev/null | python3 -c "import sys,json; [print('before', json.loads(l)['config'][:34], round(json.loads(l)['value'])) for l in sys.stdin if l.startswith('{')]" 
  cp /tmp/lib_new.so paillier_amd/libpaillier_hip.so; python tools/bench_configs.py "$@" 2>/dev/null | python3 -c "import sys,json; [print('after ', json.loads(l)['config'][:34], round(json.loads(l)['value'])) for l in sys.stdin if l.startswith('{')]" 
done
