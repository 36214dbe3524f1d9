#!/bin/bash
# slots per image and streams for 32 .. 64 images now that a level is 12 % shorter
mkdir -p gpurun_out/job32; O=gpurun_out/job32
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "$* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 32 48; do
  run --images-per-gpu $n
  for p in 7 8 12; do run --images-per-gpu $n --band-levels $p; done
  run --images-per-gpu $n --sub-batches 1
  run --images-per-gpu $n --sub-batches 3
done
run --images-per-gpu 64
for p in 6 7 8; do run --images-per-gpu 64 --update-mode 5 --band-levels $p --sub-batches 2; done
run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 4
run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 3
run --images-per-gpu 64
