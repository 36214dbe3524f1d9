#!/bin/bash
# what the (almost always empty) k_dp_sweep<UPDATE> launch behind k_band_levels costs: an experiment build without it (liblqr-hip-nosweep.so)
mkdir -p gpurun_out/job44; O=gpurun_out/job44
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
D=$PWD/gimp-lqr-plugin_amd
run() { echo -n "lib=$V $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases "$@" 2>>$O/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])"; }
for n in 8 16 48 64; do for r in 1 2; do
  V=tree run --images-per-gpu $n
  V=nosweep LQR_HIP_LIB=$D/liblqr-hip-nosweep.so run --images-per-gpu $n
done; done
