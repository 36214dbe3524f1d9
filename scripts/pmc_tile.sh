#!/bin/bash
# SQ counters for the tiled update kernel on the fhd workload (one image): where do a wave's cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tile/$tag -- python $R/bench.py --workload fhd --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_tile/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_tile"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
for f in glob.glob(root+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    if "dp_tile" in k or "vpath" in k:
        print(k, {a: round(b) for a,b in sorted(v.items())})
PY
