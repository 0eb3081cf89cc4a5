#!/bin/bash
# HBM fetch / write bytes of every long VM launch of bench.py (all configurations), one rocprofv3 --pmc pass each; run on the GPU box:
#   bash tools/bench_traffic.sh <tag>
tag=${1:-bench_traffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for set in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_$set -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_$set.log 2>&1 || exit 1
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/{tag}_{c}/**/*counter_collection.csv", recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("vm_asm") and r["Counter_Name"] == c:
                k = (int(r["Dispatch_Id"]), r["Kernel_Name"])
                per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
        out[c] = per
with open(f"gpurun_out/{tag}_summary.txt", "w") as o:
    o.write("# every VM launch of `bench.py --steps 1 --warmup 1` that fetched more than 1 GB, in launch order; bytes as the guide\n"
            "# prescribes for gfx950 (2 x FETCH_SIZE KB; WRITE_SIZE KB) -- dword gathers are not calibrated, compare launches with each other\n")
    for k in sorted(out["FETCH_SIZE"]):
        f = 2 * out["FETCH_SIZE"][k] * 1024 / 1e9; w = out["WRITE_SIZE"].get(k, 0.0) * 1024 / 1e9
        if f > 1.0:
            line = f"dispatch {k[0]:6d} {k[1]:16s} fetch {f:8.2f} GB   write {w:8.2f} GB"
            print(line); o.write(line + "\n")
PY
rm -rf gpurun_out/${tag}_FETCH_SIZE gpurun_out/${tag}_WRITE_SIZE
