#!/bin/bash
# HBM traffic of k_carve from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), 8 images per launch
tag=${1:-r01e}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pmc_$cnt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --images-per-gpu 8 > $R/gpurun_out/$tag/pmc_$cnt.log 2>&1
done
python - "$tag" <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/" + tag
out = {}
for cnt, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    vals = []
    rows = []
    for f in glob.glob(root + "/pmc_%s/**/*counter_collection.csv" % cnt, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void k_carve") and r["Counter_Name"] == cnt:
                vals.append(float(r["Counter_Value"])); rows.append(r)
    with open(root + "/pmc_%s_k_carve.csv" % cnt.lower(), "w", newline="") as g:
        if rows:
            wr = csv.DictWriter(g, fieldnames=list(rows[0].keys())); wr.writeheader(); wr.writerows(rows)
    out[key + "_size_kb_mean"] = sum(vals) / max(len(vals), 1)
    out[key + "_launches"] = len(vals)
out["images_per_launch"] = 8
out["command"] = "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --images-per-gpu 8"
out["note"] = "Counter_Value is in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section): traffic = (2*FETCH + WRITE) * 1024 bytes per launch"
json.dump(out, open(root + "/pmc_k_carve.json", "w"), indent=1)
print(out)
PY
