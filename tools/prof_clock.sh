#!/bin/bash
# effective clock per dispatch of the VM kernels: GRBM_GUI_ACTIVE (its own rocprofv3 --pmc pass) over the dispatch duration.
#   bash tools/prof_clock.sh <tag>      (on the GPU box)
set -e
tag=${1:-clock}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag -o p -- python3 bench.py --steps 3 --warmup 1 --no-traffic --no-cpu-baseline --extra-steps 1 > gpurun_out/$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
per = collections.OrderedDict()
for r in rows:
    if r["Kernel_Name"].startswith("vm_asm") and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        k = (r["Dispatch_Id"], r["Kernel_Name"][:14])
        d = per.setdefault(k, dict(cyc=0.0, s=None, e=None))
        d["cyc"] += float(r["Counter_Value"])
        for a, b in (("s", "Start_Timestamp"), ("e", "End_Timestamp")):
            if b in r: d[a] = int(r[b])
with open(f"gpurun_out/{tag}_clock.txt", "w") as o:
    for (did, nm), d in per.items():
        if d["s"] is None or d["e"] is None or d["e"] - d["s"] < 5e6: continue
        ms = (d["e"] - d["s"]) / 1e6
        line = f"{nm} dispatch {did}: {ms:8.2f} ms  GRBM_GUI_ACTIVE {d['cyc']:.4g}  -> {d['cyc'] / 8 / (ms * 1e6):.3f} GHz (cycles / 8 XCDs / duration)"
        print(line); o.write(line + "\n")
PY
rm -rf gpurun_out/$tag
