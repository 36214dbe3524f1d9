#!/bin/bash
mkdir -p gpurun_out/job12; O=gpurun_out/job12
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "tail=$LQR_TW_TAIL $* : "; timeout 600 python bench.py --steps 6 --warmup 2 --no-configs --no-cpu-baseline --no-phases "$@" 2>>$O/bench.err | python3 -c "$P"; }
for i in 1 2; do for t in 0 1; do export LQR_TW_TAIL=$t; run --images-per-gpu 64; done; done
export LQR_TW_TAIL=1
run --images-per-gpu 64 --kernel-times
LQR_TW_TAIL=0 run --images-per-gpu 64 --kernel-times
