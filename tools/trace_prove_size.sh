# kernel trace of the LAST pgpu_ddleq_prove call of tools/prove_only.py <size> [secpar]: one line per launch of >= 300 us (tools/trace_list.py)
# usage (on the GPU box): bash tools/trace_prove_size.sh 4096 [window_ms]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05; mkdir -p $o
sz=$1; win=${2:-90}
rocprofv3 --kernel-trace --output-format csv -d $o/pt_$sz -o t -- python3 tools/prove_only.py $sz > $o/pt_$sz.log 2>&1
f=$(find $o/pt_$sz -name '*kernel_trace.csv' | head -1)
python3 tools/trace_list.py "$f" $win 300 > $o/prove_trace_$sz.txt
rm -rf $o/pt_$sz
tail -3 $o/pt_$sz.log
