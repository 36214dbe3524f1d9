#!/bin/bash
# HBM traffic of the single-image kernels (k_dp_tile_p<UPDATE>, k_vpath1, k_carve at one image): the two separate PMC passes
# of profile_r03.sh on `bench.py --workload single4k`.  Writes gpurun_out/TAG/pmc_kernels_single4k.json
tag=${1:-r03c}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmcs_$cnt -- python $R/bench.py --workload single4k --steps 1 --warmup 0 --no-cpu-baseline --no-phases > $O/pmcs_$cnt.log 2>&1
done
python - "$O" <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
root = sys.argv[1]
acc = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
for cnt in acc:
    for f in glob.glob(root + "/pmcs_%s/**/*counter_collection.csv" % cnt, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cnt:
                name = r["Kernel_Name"].replace("void ", "").split("(")[0].strip()
                if name.startswith("k_"): acc[cnt][name].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --workload single4k --steps 1 --warmup 0 --no-cpu-baseline --no-phases",
       "note": "Counter_Value in KB; traffic_bytes_per_launch = (2*FETCH + WRITE) * 1024 (gfx950 correction for wide reads, MI355X_MICROARCH.md); one 3840x2160 image",
       "kernels": {}}
for name in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f, w = acc["FETCH_SIZE"].get(name, []), acc["WRITE_SIZE"].get(name, [])
    fm, wm = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    out["kernels"][name] = {"launches": max(len(f), len(w)), "fetch_size_kb_mean": round(fm, 1), "write_size_kb_mean": round(wm, 1), "traffic_bytes_per_launch": round((2 * fm + wm) * 1024)}
json.dump(out, open(root + "/pmc_kernels_single4k.json", "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:10]:
    print("%-60s launches %5d  traffic/launch %10.3f MB" % (k[:60], v["launches"], v["traffic_bytes_per_launch"] / 1e6))
PY
rm -rf $O/pmcs_FETCH_SIZE $O/pmcs_WRITE_SIZE
