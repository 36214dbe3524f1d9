#!/bin/bash
# k_band_levels at s_setprio 3 (liblqr-hip-prio.so) against the tree: 8, 16, 64 images
mkdir -p gpurun_out/job40; O=gpurun_out/job40
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
D=$PWD/gimp-lqr-plugin_amd
run() { echo -n "lib=$V $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 64 64; do for r in 1 2; do
  V=tree run --images-per-gpu $n
  V=prio LQR_HIP_LIB=$D/liblqr-hip-prio.so run --images-per-gpu $n
done; done
