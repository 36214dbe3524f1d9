#!/bin/bash
# kernel timeline of the batch step (start / end of every launch, per queue): where a sub-batch stream's seam round goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; mkdir -p $O
timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-phases "$@" > $O/bench.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$f" "$O" <<'PY'
import csv, sys, collections, gzip
f, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
keep = []
for r in rows:
    n = r["Kernel_Name"]
    if not n.startswith(("k_", "void k_")): continue
    n = n.replace("void ", "").split("(")[0].split("<")[0]
    keep.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), n))
keep.sort()
t0 = keep[0][0]
with gzip.open(out + "/timeline.csv.gz", "wt") as g:
    for s, e, q, st, n in keep: g.write("%d,%d,%s,%s,%s\n" % (s - t0, e - t0, q, st, n))
print(len(keep), "launches")
PY
rm -rf $O/prof
