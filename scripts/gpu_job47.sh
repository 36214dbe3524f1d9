#!/bin/bash
# experiment: k_band_levels on a high-priority stream of its own per sub-batch (liblqr-hip-hi.so, LQR_EXP_HI_STREAM=1), 16 hardware queues
mkdir -p gpurun_out/job47; O=gpurun_out/job47
D=$PWD/gimp-lqr-plugin_amd
export GPU_MAX_HW_QUEUES=16
run() { echo -n "hi=${LQR_EXP_HI_STREAM:-0} $* : "; LQR_HIP_LIB=$D/liblqr-hip-hi.so timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], {k: round(v['ms']/v['launches']*1000) for k,v in d['kernels_ms'].items()})"; }
for r in 1 2; do
  run --images-per-gpu 64
  LQR_EXP_HI_STREAM=1 run --images-per-gpu 64
done
run --images-per-gpu 32; LQR_EXP_HI_STREAM=1 run --images-per-gpu 32
run --images-per-gpu 48; LQR_EXP_HI_STREAM=1 run --images-per-gpu 48
tail -3 $O/bench.err
