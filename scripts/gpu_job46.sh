#!/bin/bash
# k_carve capped at 2 / 3 workgroups per CU (unused dynamic LDS) so that a level kernel's workgroup always finds room: 64 images
mkdir -p gpurun_out/job46; O=gpurun_out/job46
D=$PWD/gimp-lqr-plugin_amd
run() { echo -n "lds=${LQR_EXP_CARVE_LDS:-0} $* : "; LQR_HIP_LIB=$D/liblqr-hip-clds.so timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], {k: round(v['ms']/v['launches']*1000) for k,v in d['kernels_ms'].items()})"; }
for r in 1 2; do
  run --images-per-gpu 64
  LQR_EXP_CARVE_LDS=49152 run --images-per-gpu 64
  LQR_EXP_CARVE_LDS=65536 run --images-per-gpu 64
  LQR_EXP_CARVE_LDS=36864 run --images-per-gpu 64
done
run --images-per-gpu 16; LQR_EXP_CARVE_LDS=65536 run --images-per-gpu 16
