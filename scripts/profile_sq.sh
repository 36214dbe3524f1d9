#!/bin/bash
# SQ counters of the two chain kernels (where do the cycles of a row go): separate --pmc passes with
# --kernel-trace only, on one step of the batch (k_band_update_tw, one stream of 64 images) and of a single 4K image
# (k_dp_tile_p<UPDATE>).  Writes gpurun_out/TAG/sq_counters.json (per kernel: mean counter value per launch).
tag=${1:-r03b_sq}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_available.txt
pass=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  pass=$((pass + 1))
  for wl in batch single; do
    if [ $wl = batch ]; then args="--steps 1 --warmup 0 --sub-batches 1"; else args="--workload single4k --steps 1 --warmup 0"; fi
    timeout -s KILL 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p${pass}_$wl -- python $R/bench.py $args --no-cpu-baseline --no-phases > $O/p${pass}_$wl.log 2>&1
  done
done
python - "$O" <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/p*_*/**/*counter_collection.csv", recursive=True):
    wl = "batch" if "_batch" in f else "single"
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0].strip()
        if not (name.startswith("k_band_update_tw") or name.startswith("k_dp_tile_p") or name.startswith("k_carve") or name.startswith("k_vpath1")): continue
        if name.startswith("k_dp_tile_p") and ", true," not in name.replace("true, 1", "true,"): pass
        acc[wl + ":" + name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc <4 SQ counters per pass> --kernel-trace -- python bench.py (--sub-batches 1 | --workload single4k) --steps 1 --warmup 0",
       "note": "mean per launch, summed over the chip's SQs as rocprofv3 reports them; SQ_*_CYCLES / SQ_WAIT_* are in units of 4 cycles per the counter definitions (quad-cycles)",
       "kernels": {k: {"launches": max(len(v) for v in c.values()), **{n: round(sum(v) / len(v), 1) for n, v in sorted(c.items())}} for k, c in sorted(acc.items())}}
json.dump(out, open(root + "/sq_counters.json", "w"), indent=1)
for k, c in out["kernels"].items(): print(k[:70], {n: v for n, v in c.items()})
PY
rm -rf $O/p*_batch $O/p*_single
