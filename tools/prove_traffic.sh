#!/bin/bash
# HBM fetch / write bytes of every VM launch of the DDLEQ prover (tools/prove_only.py), one rocprofv3 --pmc pass each; run on the GPU box:
#   bash tools/prove_traffic.sh <tag> [instances] [secpar]
tag=${1:-prove_traffic}; inst=${2:-16384}; sp=${3:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for set in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_$set -o p -- python3 tools/prove_only.py $inst $sp > gpurun_out/${tag}_$set.log 2>&1 || exit 1
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
ks = sorted(out["FETCH_SIZE"])
with open(f"gpurun_out/{tag}_summary.txt", "w") as o:
    for k in ks[-16:]:
        f = out["FETCH_SIZE"][k]; w = out["WRITE_SIZE"].get(k, 0.0)
        # gfx950: FETCH_SIZE counts 64-byte units as kilobytes / 2 -> bytes = 2 x value x 1024 ... as the guide prescribes (x 1024 for KB, x 2)
        line = f"dispatch {k[0]:5d} {k[1]:16s} fetch {2 * f * 1024 / 1e9:8.2f} GB   write {w * 1024 / 1e9:8.2f} GB"
        print(line); o.write(line + "\n")
PY
