#!/bin/bash
# A/B of the band-update kernels: parity subset per variant, then kernel times at 64 x 4K
#   scripts/exp_band.sh TAG "variants"
tag=${1:-exp}; vars=${2:-"0 1 2"}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for v in $vars; do
  if [ $v != 0 ]; then
    LQR_BAND_KERNEL=$v timeout 600 python -m pytest tests/test_round2_gpu.py tests/test_parity_gpu.py tests/test_tolerance_boundary.py -m gpu -x -q -k "band and not mw or planes or carve_sides" > $O/parity_v$v.log 2>&1
    tail -3 $O/parity_v$v.log
  fi
  timeout 300 python bench.py --band-kernel $v --kernel-times --seams 60 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_v$v.json 2> $O/bench_v$v.err
  python - $O/bench_v$v.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print("value", j["value"], "ms/step", j["ms_per_step"], {k:(round(v["ms"]/v["launches"],4)) for k,v in j["kernels_ms"].items()})
except Exception as e: print("bench failed", e)
PY
done
