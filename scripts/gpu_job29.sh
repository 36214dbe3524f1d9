#!/bin/bash
# k_dp_tile_p: XCD-contiguous tiles with (0) and without (1) the near copies, alternating
mkdir -p gpurun_out/job29; O=gpurun_out/job29
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "DPP_DBG=${LQR_DPP_DBG:-0} $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for d in 0 1 0 1 0 1; do LQR_DPP_DBG=$d run --workload single4k; done
for d in 0 1 0 1; do LQR_DPP_DBG=$d run --workload config5; done
for d in 0 1 0 1; do LQR_DPP_DBG=$d run --images-per-gpu 4; done
for d in 0 1 0 1; do LQR_DPP_DBG=$d run --images-per-gpu 2; done
