#!/bin/bash
# 64 images on more streams (the chain, not the chip, bounds 4 streams of 16: 96 images run 621 k)
mkdir -p gpurun_out/job39; O=gpurun_out/job39
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "Q=${GPU_MAX_HW_QUEUES:-8} $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
run --images-per-gpu 64
run --images-per-gpu 64 --sub-batches 6
run --images-per-gpu 64 --sub-batches 8
GPU_MAX_HW_QUEUES=16 run --images-per-gpu 64 --sub-batches 8
run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 8
run --images-per-gpu 64 --update-mode 5 --band-levels 10 --sub-batches 8
GPU_MAX_HW_QUEUES=16 run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 8
GPU_MAX_HW_QUEUES=16 run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 6
GPU_MAX_HW_QUEUES=16 run --images-per-gpu 64 --sub-batches 4
