#!/bin/bash
mkdir -p gpurun_out/job3; O=gpurun_out/job3
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()}, d.get("band_levels_stats"), d.get("band_tiles_stats"), d["roofline"]["frac"], d["roofline"].get("moved_bytes_per_launch"), d["roofline"]["alg_bytes_per_launch"])'
run() { echo -n "$* : "; timeout 300 python bench.py --steps 4 --warmup 2 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
run --images-per-gpu 64
run --images-per-gpu 64 --update-mode 5 --band-levels 6
run --images-per-gpu 64 --update-mode 5 --band-levels 8
run --images-per-gpu 64 --update-mode 5 --band-levels 5
run --images-per-gpu 64 --update-mode 5 --band-levels 6 --sub-batches 2
run --images-per-gpu 16
run --images-per-gpu 16 --update-mode 5 --band-levels 8
run --images-per-gpu 16 --update-mode 5 --band-levels 12
run --images-per-gpu 8
run --images-per-gpu 8 --update-mode 5 --band-levels 12
run --images-per-gpu 8 --update-mode 5 --band-levels 16
run --images-per-gpu 1 --update-mode 5 --band-levels 16
run --images-per-gpu 1
tail -5 $O/bench.err
