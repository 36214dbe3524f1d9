#!/bin/bash
# Round-2 evidence for profiles/: bench line + rocprofv3 kernel stats of the same command, k_carve's HBM traffic
# from two separate PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains), and the
# single-image workloads (configs 2, 3, 5 geometry) with their kernel stats.
#   scripts/profile_r02.sh TAG [what...]      what: batch pmc single (default: all)
tag=${1:-r02x}; shift
what=${*:-batch pmc single}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
stats() {   # stats NAME bench-args...
  name=$1; shift
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python $R/bench.py --no-cpu-baseline "$@" > $O/${name}_prof.log 2>&1
  find $O/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  grep '^{' $O/${name}_prof.log | tail -1 > $O/${name}_bench_under_rocprof.json
  rm -rf $O/prof_$name
  head -14 $O/${name}_kernel_stats.csv
}
for w in $what; do
case $w in
batch)
  timeout -s KILL 400 python $R/bench.py --steps 5 --warmup 2 > $O/batch4k_bench.json 2> $O/batch4k_bench.err
  cut -c1-1500 $O/batch4k_bench.json
  stats batch4k --steps 3 --warmup 1 ;;
pmc)
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_$cnt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$cnt.log 2>&1
  done
  python - "$O" <<'PY'
import csv, glob, json, sys
root = sys.argv[1]
out = {}
for cnt, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    vals, rows = [], []
    for f in glob.glob(root + "/pmc_%s/**/*counter_collection.csv" % cnt, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_carve" in r["Kernel_Name"] and r["Counter_Name"] == cnt:
                vals.append(float(r["Counter_Value"])); rows.append(r)
    with open(root + "/pmc_%s_k_carve.csv" % cnt.lower(), "w", newline="") as g:
        if rows:
            wr = csv.DictWriter(g, fieldnames=list(rows[0].keys())); wr.writeheader(); wr.writerows(rows)
    out[key + "_size_kb_mean"] = sum(vals) / max(len(vals), 1)
    out[key + "_launches"] = len(vals)
out["images_per_launch"] = 64
out["command"] = "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline"
out["note"] = ("Counter_Value is in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section): "
               "traffic = (2*FETCH + WRITE) * 1024 bytes per launch")
json.dump(out, open(root + "/pmc_k_carve.json", "w"), indent=1)
print(out)
PY
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
single)
  for wl in fhd single4k 8k; do
    timeout -s KILL 300 python $R/bench.py --workload $wl --steps 3 --warmup 1 > $O/${wl}_bench.json 2> $O/${wl}_bench.err
    cut -c1-600 $O/${wl}_bench.json
    stats $wl --workload $wl --steps 2 --warmup 1
  done ;;
esac
done
