#!/bin/bash
mkdir -p gpurun_out/job5; O=gpurun_out/job5
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()}, d.get("band_levels_stats"))'
run() { echo -n "dbg=$LQR_LV_DBG $* : "; timeout 300 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for d in 0 1 2 3; do export LQR_LV_DBG=$d
run --images-per-gpu 16 --update-mode 5 --band-levels 12
run --images-per-gpu 1 --update-mode 5 --band-levels 16
done
