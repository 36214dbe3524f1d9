#!/bin/bash
# throughput of the 64 x 4K batch by number of sub-batch streams and hardware queues
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
for cfg in "8 2" "8 4" "12 8" "20 8" "20 16" "24 16"; do
  set -- $1 $cfg
  q=$2; n=$3
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --sub-batches $n --steps 3 --warmup 1 --no-cpu-baseline --no-phases > gpurun_out/$1/s${n}_q$q.json 2> gpurun_out/$1/err.log
  python - gpurun_out/$1/s${n}_q$q.json $q $n <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); print("queues", sys.argv[2], "streams asked", sys.argv[3], "used", j["config"]["streams_per_gpu"], "value", j["value"], "ms/step", j["ms_per_step"], "carve frac", j["roofline"]["frac"])
except Exception as e: print("failed", sys.argv[2:], e)
PY
done
