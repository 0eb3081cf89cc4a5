#!/bin/bash
# A/B of two builds on one box, alternating: the tree's library against paillier_amd/libpaillier_hip_ab.so (PGPU_LIBRARY).
#   bash tools/ab_lib.sh "<prove_only args>" [rounds]
args=${1:-16384}; rounds=${2:-3}
for r in $(seq $rounds); do
  for lib in tree ab; do
    if [ $lib = ab ]; then export PGPU_LIBRARY=$GRAFT_REPO_ROOT/paillier_amd/libpaillier_hip_ab.so; else unset PGPU_LIBRARY; fi
    echo "$lib: $(PROVE_REPS=4 python3 tools/prove_only.py $args 2>&1 | grep '^prove' | awk '{printf "%.1f ", $5}')"
  done
done
