#!/bin/bash
# the word store of a level delayed by s_sleep 1 / 2 / 4 after the granule stores (liblqr-hip-s{1,2,4}.so) against the tree
mkdir -p gpurun_out/job36; O=gpurun_out/job36
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
D=$PWD/gimp-lqr-plugin_amd
run() { echo -n "lib=$V $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 32; do for r in 1 2; do
  V=tree run --images-per-gpu $n
  for s in 1 2 4; do V=s$s LQR_HIP_LIB=$D/liblqr-hip-s$s.so run --images-per-gpu $n; done
done; done
