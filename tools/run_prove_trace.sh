cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05; mkdir -p $o
for a in "16384" "8192"; do echo "== $a"; PGPU_PROFILE_DUMP=1 PGPU_HOST_TRACE=1 PROVE_REPS=3 python3 tools/prove_only.py $a 2>&1 | grep -v amdgpu.ids | tail -13 | cut -c1-360; done > gpurun_out/r5_prove7.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $o/prove_kt2 -o t -- python3 tools/prove_only.py 16384 > $o/prove_kt2.log 2>&1
f=$(find $o/prove_kt2 -name '*kernel_trace.csv' | head -1); cp "$f" $o/prove_trace2.csv; rm -rf $o/prove_kt2
