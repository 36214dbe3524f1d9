#!/bin/bash
# exp_timing.sh TAG "variants" "nimgs" seams
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
for n in ${3:-64}; do for v in $2; do timeout 300 python scripts/exp_band_timing.py $v $n ${4:-40} 2>&1 | grep -v amdgpu.ids | tail -20 | tee gpurun_out/$1/timing_v${v}_n$n.txt; done; done
