#!/bin/bash
# kernel timeline of a short pipelined run: do kernels of the two sub-batches overlap in time?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/trace
LQRHIP_PIPE=${PIPE:-1} LQRHIP_SUBBATCHES=2 timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace/p -- python $R/bench.py --steps 1 --warmup 0 --seams 12 --no-cpu-baseline ${EXTRA} > $R/gpurun_out/trace/log.txt 2>&1
tail -1 $R/gpurun_out/trace/log.txt | cut -c1-300
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace/p/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_carve", "k_band_update", "k_vpath", "k_emap_update"))]
for r in sel[60:110]:
    print("%-28s q=%s  start %9.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0][:28], r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
