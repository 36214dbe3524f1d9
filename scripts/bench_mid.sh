#!/bin/bash
pp='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:(round(v["ms"],1),v["launches"]) for k,v in d["kernels_ms"].items()})'
for n in "$@"; do for px in 0 1000000000; do echo "n=$n tiled_px=$px"; LQRHIP_TILED_UPDATE_PX=$px timeout -s KILL 120 python bench.py --images-per-gpu $n --steps 2 --warmup 1 --no-cpu-baseline --kernel-times 2>&1 | tail -1 | python -c "$pp"; done; done
