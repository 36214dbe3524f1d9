#!/bin/bash
mkdir -p gpurun_out/job10; O=gpurun_out/job10
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()}, d.get("band_levels_stats"))'
run() { echo -n "$* : "; timeout 300 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 12 16 24 32 40; do run --images-per-gpu $n; done
for n in 24 32 40 48; do run --images-per-gpu $n --update-mode 0; done
run --images-per-gpu 48 --update-mode 5
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/suite.log 2>&1; echo "suite rc $?"; tail -3 $O/suite.log
