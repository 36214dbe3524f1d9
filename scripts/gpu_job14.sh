#!/bin/bash
mkdir -p gpurun_out/job14; O=gpurun_out/job14
timeout 600 python scripts/gpu_levels_quick.py > $O/levels_quick.log 2>&1; echo "levels quick rc $?"; grep -c "^ok" $O/levels_quick.log; grep "^FAIL" $O/levels_quick.log | head -5
FUZZ_LEVELS=1 FUZZ_COUNT=150 timeout 900 python scripts/fuzz_tiles.py 0 5511 > $O/fuzz_levels.log 2>&1; echo "fuzz levels rc $?"; tail -1 $O/fuzz_levels.log
FUZZ_COUNT=100 GPU_MAX_HW_QUEUES=8 timeout 900 python scripts/fuzz_batch.py 0 5512 > $O/fuzz_batch.log 2>&1; echo "fuzz batch rc $?"; tail -1 $O/fuzz_batch.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "$* : "; timeout 600 python bench.py --steps 4 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
run --images-per-gpu 1 --update-mode 5 --band-levels 16
run --images-per-gpu 8
run --images-per-gpu 16
run --images-per-gpu 32
run --images-per-gpu 48
run --images-per-gpu 64 --update-mode 5 --band-levels 7
run --images-per-gpu 64
