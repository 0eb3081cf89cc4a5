#!/bin/bash
# Round-4 numbers from ONE box, after the last commit that touches paillier_amd/csrc (run on the GPU box): the bench line, kernel trace /
# headline stats / PMC summary of the same command, the prover's launch lists, the per-rank floors.  Everything lands in gpurun_out/r04/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04; mkdir -p $o
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_line.json 2> $o/bench_line.err
python3 tools/bench_summary.py $o/bench_line.json | tee $o/bench_summary.txt
bash tools/prof_r04.sh kernel-trace > $o/prof_kt.txt 2>&1; cp gpurun_out/prof_r04/bench_kernel_stats.csv $o/ || true
bash tools/prof_r04.sh headline > $o/prof_headline.txt 2>&1; cp gpurun_out/prof_r04/headline_kernel_stats.csv $o/ || true
tail -2 $o/prof_headline.txt | cut -c1-300
for a in "16384" "61440 40" "2048" "4096" "8192"; do
  echo "== prove_only.py $a" >> $o/prove_launches_raw.txt
  PGPU_PROFILE_DUMP=1 PGPU_HOST_TRACE=1 PROVE_REPS=3 PROVE_VERIFY=1 python3 tools/prove_only.py $a 2>&1 | grep -v "amdgpu.ids" | tail -16 | cut -c1-260 >> $o/prove_launches_raw.txt
done
python3 tools/rank_floor.py 3 > $o/rank_floor.json 2> /dev/null
python3 tools/threshold_shard_probe.py > $o/threshold_shard_probe.jsonl 2> /dev/null
echo done
