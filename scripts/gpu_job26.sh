#!/bin/bash
# the L2-resident second copy of the level hand-off (LQR_LV_DBG=4) against the default, 8..64 images, after a parity check
mkdir -p gpurun_out/job26; O=gpurun_out/job26
LQR_LV_DBG=4 timeout 600 python scripts/gpu_levels_quick.py > $O/quick_dbg4.log 2>&1; echo "quick dbg4 exit $?"; tail -3 $O/quick_dbg4.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "LV_DBG=${LQR_LV_DBG:-0} $* : "; timeout 600 python bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 16 48; do
  for d in 0 4 0 4; do LQR_LV_DBG=$d run --images-per-gpu $n; done
done
for d in 0 4; do LQR_LV_DBG=$d run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 2; done
