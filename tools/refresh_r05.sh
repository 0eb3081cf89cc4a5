#!/bin/bash
# Round-5 measurements from ONE box (run on the GPU box): prover kernel trace + launch list, small-batch sweep, per-rank floors.
# Everything lands in gpurun_out/r05/.    bash tools/refresh_r05.sh [trace|sweep|floor|launches ...]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05; mkdir -p $o
what=${@:-bench trace sweep floor launches}
for w in $what; do
  case $w in
    bench)
      python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_line.json 2> $o/bench_line.err
      python3 tools/bench_summary.py $o/bench_line.json | tee $o/bench_summary.txt;;
    kt) bash tools/prof_r05.sh kernel-trace > $o/prof_kt.txt 2>&1; cp gpurun_out/prof_r05/bench_kernel_stats.csv $o/ || true;;
    headline) bash tools/prof_r05.sh headline > $o/prof_headline.txt 2>&1; cp gpurun_out/prof_r05/headline_kernel_stats.csv $o/ || true; tail -2 $o/prof_headline.txt | cut -c1-300;;
    pmc) bash tools/prof_r05.sh pmc > $o/prof_pmc.txt 2>&1; cp gpurun_out/prof_r05/bench_pmc_summary.txt $o/ || true;;
    trace)
      rocprofv3 --kernel-trace --output-format csv -d $o/prove_kt -o t -- python3 tools/prove_only.py 16384 > $o/prove_kt.log 2>&1
      f=$(find $o/prove_kt -name '*kernel_trace.csv' | head -1)
      cp "$f" $o/prove_trace.csv; rm -rf $o/prove_kt
      python3 tools/trace_gaps.py $o/prove_trace.csv 135 > $o/prove_gaps.txt; tail -3 $o/prove_kt.log;;
    sweep) python3 tools/small_batch_sweep.py 2048 4096 > $o/small_batch_sweep.jsonl 2> $o/small_batch_sweep.err; tail -3 $o/small_batch_sweep.jsonl | cut -c1-200;;
    floor) python3 tools/rank_floor.py 3 > $o/rank_floor.json 2> /dev/null; head -c 600 $o/rank_floor.json;;
    launches)
      : > $o/prove_launches_raw.txt
      for a in "16384" "61440 40" "2048" "4096" "8192"; do
        echo "== prove_only.py $a" >> $o/prove_launches_raw.txt
        PGPU_PROFILE_DUMP=1 PGPU_HOST_TRACE=1 PROVE_REPS=3 PROVE_VERIFY=1 python3 tools/prove_only.py $a 2>&1 | grep -v "amdgpu.ids" | tail -16 | cut -c1-260 >> $o/prove_launches_raw.txt
      done;;
  esac
done
echo done
