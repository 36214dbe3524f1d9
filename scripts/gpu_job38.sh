#!/bin/bash
# 64 images: k_band_update_tw (with scalar plane bases) against k_band_levels on 4 streams, alternating; 56 and 96 images too
mkdir -p gpurun_out/job38; O=gpurun_out/job38
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "$* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for r in 1 2 3 4; do run --images-per-gpu 64; run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 4; done
run --images-per-gpu 56; run --images-per-gpu 56 --update-mode 5 --band-levels 7 --sub-batches 4
run --images-per-gpu 96; run --images-per-gpu 96 --update-mode 5 --band-levels 7 --sub-batches 4
