#!/bin/bash
# after trimming the level loop's LDS reads: parity, then 8 .. 64 images; 64 images alternating _tw (default) and levels on 4 streams
mkdir -p gpurun_out/job33; O=gpurun_out/job33
timeout 600 python scripts/gpu_levels_quick.py > $O/quick.log 2>&1; echo "quick exit $?"; tail -1 $O/quick.log
timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests exit $?"; tail -2 $O/tests.log
FUZZ_COUNT=300 FUZZ_LEVELS=1 timeout 1200 python scripts/fuzz_tiles.py 0 81 > $O/fuzz_tiles_levels.log 2>&1; echo "fuzz_tiles exit $?"; tail -1 $O/fuzz_tiles_levels.log
FUZZ_COUNT=40 timeout 1200 python scripts/fuzz_batch.py 0 82 > $O/fuzz_batch.log 2>&1; echo "fuzz_batch exit $?"; tail -1 $O/fuzz_batch.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "$* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 16 32 40 48; do run --images-per-gpu $n; done
for i in 1 2 3; do
  run --images-per-gpu 64
  run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 4
  run --images-per-gpu 64 --update-mode 5 --band-levels 7 --sub-batches 2
done
