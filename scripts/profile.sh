#!/bin/bash
# Evidence for profiles/, all from ONE commit: the driver's bench line (with configs 2/3/5 and the per-kernel
# breakdown on it), rocprofv3 --kernel-trace --stats of the same command and of the single-image workloads, and the HBM
# traffic of every kernel from two SEPARATE PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains)
# for the batch and for the single 4K image -> pmc_kernels.json, which bench.py reads for roofline.traffic / pmc_bytes.
#   scripts/profile.sh TAG [what...]      what: batch stats pmc (default: all)
tag=${1:-r06x}; shift
what=${*:-batch stats pmc}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
stats() {   # stats NAME bench-args...
  name=$1; shift
  timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $R/bench.py --no-cpu-baseline --no-phases --no-configs --no-kernel-breakdown "$@" > $O/${name}_prof.log 2>&1
  find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  grep '^{' $O/${name}_prof.log | tail -1 > $O/${name}_bench_under_rocprof.json
  rm -rf /tmp/prof_$name $O/${name}_prof.log
  head -12 $O/${name}_kernel_stats.csv | cut -c1-160
}
pmc() {     # pmc NAME images_per_launch bench-args...
  name=$1; ipl=$2; shift; shift
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 500 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d /tmp/pmc_${name}_$cnt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-phases --no-configs --no-kernel-breakdown "$@" > /tmp/pmc_${name}_$cnt.log 2>&1
  done
  python - "$name" "$ipl" "$O" "$*" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
name, ipl, out_dir, args = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
acc = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
for cnt in acc:
    for f in glob.glob("/tmp/pmc_%s_%s/**/*counter_collection.csv" % (name, cnt), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cnt:
                k = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "")).split("(")[0].strip()
                if k.startswith("k_"):
                    acc[cnt][k].append(float(r["Counter_Value"]))
entry = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-phases --no-configs --no-kernel-breakdown " + args,
         "note": "Counter_Value is in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section): traffic_bytes_per_launch = (2*FETCH + WRITE) * 1024",
         "images_per_launch": ipl, "kernels": {}}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f, w = acc["FETCH_SIZE"].get(k, []), acc["WRITE_SIZE"].get(k, [])
    fm, wm = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    entry["kernels"][k] = {"launches": max(len(f), len(w)), "fetch_size_kb_mean": round(fm, 1), "write_size_kb_mean": round(wm, 1), "traffic_bytes_per_launch": round((2 * fm + wm) * 1024)}
path = os.path.join(out_dir, "pmc_kernels.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
allw[name] = entry
json.dump(allw, open(path, "w"), indent=1)
for k, v in sorted(entry["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:10]:
    print("%-10s %-24s launches %5d  traffic/launch %10.3f MB" % (name, k[:24], v["launches"], v["traffic_bytes_per_launch"] / 1e6))
PY
  rm -rf /tmp/pmc_${name}_FETCH_SIZE /tmp/pmc_${name}_WRITE_SIZE
}
for w in $what; do
case $w in
batch)
  timeout -s KILL 900 python $R/bench.py --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2> $O/driver_cmd_bench.err
  cut -c1-1200 $O/driver_cmd_bench.json; echo
  timeout -s KILL 400 python $R/bench.py --steps 5 --warmup 2 --sub-batches 1 --no-cpu-baseline --no-phases --no-configs > $O/batch4k_one_stream_bench.json 2>> $O/driver_cmd_bench.err
  timeout -s KILL 400 python $R/bench.py --steps 5 --warmup 2 --images-per-gpu 8 --no-cpu-baseline --no-phases --no-configs > $O/batch4k_8img_bench.json 2>> $O/driver_cmd_bench.err
  timeout -s KILL 400 python $R/bench.py --steps 5 --warmup 2 --images-per-gpu 16 --no-cpu-baseline --no-phases --no-configs > $O/batch4k_16img_bench.json 2>> $O/driver_cmd_bench.err
  timeout -s KILL 600 python $R/bench.py --steps 2 --warmup 1 --images-per-gpu 16 --delta 2 --no-cpu-baseline --no-phases --no-configs > $O/batch4k_16img_delta2_bench.json 2>> $O/driver_cmd_bench.err
  timeout -s KILL 600 python $R/bench.py --steps 2 --warmup 1 --images-per-gpu 16 --delta 3 --no-cpu-baseline --no-phases --no-configs > $O/batch4k_16img_delta3_bench.json 2>> $O/driver_cmd_bench.err
  timeout -s KILL 400 python $R/bench.py --steps 5 --warmup 2 --images-per-gpu 32 --no-cpu-baseline --no-phases --no-configs > $O/batch4k_32img_bench.json 2>> $O/driver_cmd_bench.err
  timeout -s KILL 600 python $R/bench.py --workload config5 --delta 2 --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $O/config5_delta2_bench.json 2>> $O/driver_cmd_bench.err
  timeout -s KILL 600 python $R/bench.py --workload config5 --rigmask --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $O/config5_rigmask_bench.json 2>> $O/driver_cmd_bench.err
  for f in batch4k_one_stream batch4k_8img batch4k_16img batch4k_32img batch4k_16img_delta2 batch4k_16img_delta3 config5_delta2 config5_rigmask; do python -c "
import json; d=json.load(open('$O/${f}_bench.json')); print('$f', d['value'], d['ms_per_step'])"; done ;;
stats)
  stats batch4k --steps 3 --warmup 1
  stats batch4k_8img --steps 3 --warmup 1 --images-per-gpu 8
  stats batch4k_16img --steps 3 --warmup 1 --images-per-gpu 16
  stats fhd --workload fhd --steps 2 --warmup 1
  stats single4k --workload single4k --steps 2 --warmup 1
  stats config5 --workload config5 --steps 1 --warmup 0 ;;
pmc)
  pmc batch4k 16
  pmc batch4k_8img 8 --images-per-gpu 8
  pmc single4k 1 --workload single4k
  pmc fhd 1 --workload fhd
  pmc config5 1 --workload config5 ;;
esac
done
