#!/bin/bash
# exp_bench.sh TAG "band kernels" [extra bench args]: kernel times at 64 x 4K per band-kernel variant
tag=${1:-exp}; vars=${2:-"0"}; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for v in $vars; do
  timeout 300 python bench.py --band-kernel $v --kernel-times --seams 60 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/bench_v$v.json 2> $O/bench_v$v.err
  python - $O/bench_v$v.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print("value", j["value"], "ms/step", j["ms_per_step"], {k:(round(v["ms"]/v["launches"],4)) for k,v in j["kernels_ms"].items()})
except Exception as e: print("bench failed", e)
PY
done
