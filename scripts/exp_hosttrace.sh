#!/bin/bash
# host-side cost of the seam loop: HIP runtime API trace of one bench run (no counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hosttrace; mkdir -p $O
timeout -s KILL 500 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-phases "$@" > $O/bench.log 2>&1
f=$(find $O/prof -name "*hip_api_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
d = collections.defaultdict(list)
for r in rows: d[r["Function"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-36s n=%6d total %.1f ms mean %.1f us median %.1f us" % (k, len(v), sum(v) / 1e6, st.mean(v) / 1e3, st.median(v) / 1e3))
# launch cadence of the main thread in the last step: time between consecutive hipLaunchKernel starts
ls = sorted(int(r["Start_Timestamp"]) for r in rows if "Launch" in r["Function"])
ls = ls[-4000:]
gaps = [b - a for a, b in zip(ls[:-1], ls[1:])]
print("launch-to-launch: mean %.1f us median %.1f us p90 %.1f us" % (st.mean(gaps) / 1e3, st.median(gaps) / 1e3, sorted(gaps)[int(len(gaps) * 0.9)] / 1e3))
PY
grep '^{' $O/bench.log | tail -1 | cut -c1-200
rm -rf $O/prof
