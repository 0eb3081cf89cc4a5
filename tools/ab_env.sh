#!/bin/bash
# A/B of an environment switch on one box, alternating:  bash tools/ab_env.sh VAR "<prove_only args>" [rounds]
var=$1; args=${2:-16384}; rounds=${3:-3}
for r in $(seq $rounds); do
  for v in 0 1; do
    echo "$var=$v: $(env $var=$v PROVE_REPS=5 python3 tools/prove_only.py $args 2>&1 | grep '^prove' | awk '{printf "%.1f ", $5}')"
  done
done
