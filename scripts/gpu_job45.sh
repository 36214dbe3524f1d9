#!/bin/bash
# k_carve with groups of 2 and 3 chunks (74 / 86 VGPRs: more waves beside a level kernel's wave) against the tree's 4 (96 VGPRs)
mkdir -p gpurun_out/job45; O=gpurun_out/job45
D=$PWD/gimp-lqr-plugin_amd
run() { echo -n "lib=$V $* : "; timeout 600 python bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-phases "$@" 2>>$O/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), d['ms_per_step'], 'carve', r['avg_launch_us'], r['frac'], 'alone', (r.get('alone') or {}).get('frac'))"; }
for n in 64 8 96; do for r in 1 2; do
  V=tree run --images-per-gpu $n
  V=cg2 LQR_HIP_LIB=$D/liblqr-hip-cg2.so run --images-per-gpu $n
  V=cg3 LQR_HIP_LIB=$D/liblqr-hip-cg3.so run --images-per-gpu $n
done; done
