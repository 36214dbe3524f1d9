#!/bin/bash
# Round-3 evidence for profiles/: bench lines + rocprofv3 kernel stats of the same commands, HBM traffic of EVERY kernel of
# the batch step from two separate PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains), the
# single-image workloads (configs 2, 3) and config 5 as stated with its two variants.
#   scripts/profile_r03.sh TAG [what...]      what: batch pmc single config5 (default: all)
tag=${1:-r03x}; shift
what=${*:-batch pmc single config5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
stats() {   # stats NAME bench-args...
  name=$1; shift
  timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python $R/bench.py --no-cpu-baseline --no-phases "$@" > $O/${name}_prof.log 2>&1
  find $O/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  grep '^{' $O/${name}_prof.log | tail -1 > $O/${name}_bench_under_rocprof.json
  rm -rf $O/prof_$name
  head -16 $O/${name}_kernel_stats.csv
}
for w in $what; do
case $w in
batch)
  timeout -s KILL 600 python $R/bench.py --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2> $O/driver_cmd_bench.err
  cut -c1-2500 $O/driver_cmd_bench.json
  timeout -s KILL 400 python $R/bench.py --steps 5 --warmup 2 --sub-batches 1 --no-cpu-baseline --no-phases > $O/batch4k_one_stream_bench.json 2>> $O/driver_cmd_bench.err
  cut -c1-400 $O/batch4k_one_stream_bench.json
  timeout -s KILL 400 python $R/bench.py --steps 5 --warmup 2 --images-per-gpu 8 --no-cpu-baseline --no-phases > $O/batch4k_8img_bench.json 2>> $O/driver_cmd_bench.err
  cut -c1-400 $O/batch4k_8img_bench.json
  stats batch4k --steps 3 --warmup 1 ;;
pmc)
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 400 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_$cnt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-phases > $O/pmc_$cnt.log 2>&1
  done
  python - "$O" <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
root = sys.argv[1]
acc = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
for cnt in acc:
    for f in glob.glob(root + "/pmc_%s/**/*counter_collection.csv" % cnt, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cnt:
                name = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "")).split("(")[0].strip()
                acc[cnt][name].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-phases",
       "note": ("Counter_Value is in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section): "
                "traffic_bytes_per_launch = (2*FETCH + WRITE) * 1024.  The batch runs as 4 sub-batches of 16 images, one launch per sub-batch."),
       "kernels": {}}
for name in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f, w = acc["FETCH_SIZE"].get(name, []), acc["WRITE_SIZE"].get(name, [])
    fm, wm = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    out["kernels"][name] = {"launches": max(len(f), len(w)), "fetch_size_kb_mean": round(fm, 1), "write_size_kb_mean": round(wm, 1),
                            "traffic_bytes_per_launch": round((2 * fm + wm) * 1024)}
json.dump(out, open(root + "/pmc_kernels.json", "w"), indent=1)
kc = out["kernels"].get("k_carve")
if kc:
    json.dump({"fetch_size_kb_mean": kc["fetch_size_kb_mean"], "write_size_kb_mean": kc["write_size_kb_mean"], "fetch_launches": kc["launches"],
               "write_launches": kc["launches"], "images_per_launch": 16, "command": out["command"], "note": out["note"]}, open(root + "/pmc_k_carve.json", "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:14]:
    print("%-28s launches %5d  traffic/launch %10.3f MB" % (k[:28], v["launches"], v["traffic_bytes_per_launch"] / 1e6))
PY
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
single)
  for wl in fhd single4k; do
    timeout -s KILL 300 python $R/bench.py --workload $wl --steps 3 --warmup 1 > $O/${wl}_bench.json 2> $O/${wl}_bench.err
    cut -c1-500 $O/${wl}_bench.json
    stats $wl --workload $wl --steps 2 --warmup 1
  done ;;
config5)
  timeout -s KILL 600 python $R/bench.py --workload config5 --steps 2 --warmup 1 > $O/config5_bench.json 2> $O/config5_bench.err
  cut -c1-500 $O/config5_bench.json
  timeout -s KILL 600 python $R/bench.py --workload config5 --delta 2 --steps 1 --warmup 1 --no-cpu-baseline > $O/config5_delta2_bench.json 2>> $O/config5_bench.err
  cut -c1-500 $O/config5_delta2_bench.json
  timeout -s KILL 600 python $R/bench.py --workload config5 --rigmask --steps 1 --warmup 1 --no-cpu-baseline > $O/config5_rigmask_bench.json 2>> $O/config5_bench.err
  cut -c1-500 $O/config5_rigmask_bench.json
  stats config5 --workload config5 --steps 1 --warmup 0 ;;
esac
done
