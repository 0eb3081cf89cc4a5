#!/bin/bash
# A/B of the placement by LDS size (library flag "exclusive": a CU per small workgroup, the main ladder spread over the CUs): ms per
# prove call at small batches (tools/prove_only.py)
set -e
for n in 2048 4096 8192; do
  for f in 0 1; do
    echo "instances $n exclusive $f: $(PROVE_EXCLUSIVE=$f PROVE_REPS=4 PROVE_VERIFY=1 python3 tools/prove_only.py $n 2>&1 | grep '^prove\|^verify' | awk '{ if ($1=="prove") printf "%.1f ", $5; else printf "| verify accepted %s of %s", $6, $8 }')"
  done
done
