#!/bin/bash
mkdir -p gpurun_out/job25; O=gpurun_out/job25
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "$* : "; timeout 600 python bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
run --images-per-gpu 64
run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 2
run --images-per-gpu 64 --update-mode 5 --band-levels 8 --sub-batches 2
run --images-per-gpu 64 --update-mode 5 --band-levels 6 --sub-batches 2
run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 3
run --images-per-gpu 64 --update-mode 5 --band-levels 10 --sub-batches 2
run --images-per-gpu 64
run --images-per-gpu 64 --sub-batches 3
