#!/bin/bash
mkdir -p gpurun_out/job24; O=gpurun_out/job24
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "div=$LQRHIP_CARVE_DIV lib=${LQR_HIP_LIB##*/} $* : "; timeout 600 python bench.py --steps 6 --warmup 2 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for d in 1 2 4 8 1 4; do export LQRHIP_CARVE_DIV=$d; run --images-per-gpu 64; done
unset LQRHIP_CARVE_DIV
export LQR_HIP_LIB=$PWD/gimp-lqr-plugin_amd/liblqr-hip-x.so; run --images-per-gpu 64; unset LQR_HIP_LIB; run --images-per-gpu 64
export LQR_HIP_LIB=$PWD/gimp-lqr-plugin_amd/liblqr-hip-x.so; run --images-per-gpu 64; run --images-per-gpu 16; unset LQR_HIP_LIB; run --images-per-gpu 16
run --images-per-gpu 32 --sub-batches 2
run --images-per-gpu 32
run --images-per-gpu 48 --sub-batches 2
run --images-per-gpu 48
