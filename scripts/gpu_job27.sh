#!/bin/bash
# the near copies of k_band_levels' hand-over as the default: parity (quick, the new placement test, count-bounded fuzz with
# levels forced and the batch fuzz), then cross-XCD placement against the default for speed
mkdir -p gpurun_out/job27; O=gpurun_out/job27
timeout 600 python scripts/gpu_levels_quick.py > $O/quick.log 2>&1; echo "quick exit $?"; tail -1 $O/quick.log
timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -x -q > $O/round5.log 2>&1; echo "round5 exit $?"; tail -2 $O/round5.log
FUZZ_COUNT=400 FUZZ_LEVELS=1 timeout 1200 python scripts/fuzz_tiles.py 0 51 > $O/fuzz_tiles_levels.log 2>&1; echo "fuzz_tiles exit $?"; tail -1 $O/fuzz_tiles_levels.log
FUZZ_COUNT=60 timeout 1200 python scripts/fuzz_batch.py 0 52 > $O/fuzz_batch.log 2>&1; echo "fuzz_batch exit $?"; tail -1 $O/fuzz_batch.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "LV_DBG=${LQR_LV_DBG:-0} $* : "; timeout 600 python bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 32; do for d in 0 4 8; do LQR_LV_DBG=$d run --images-per-gpu $n; done; done
