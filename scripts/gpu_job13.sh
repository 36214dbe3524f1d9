#!/bin/bash
mkdir -p gpurun_out/job13; O=gpurun_out/job13
timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -q -x -p no:cacheprovider -k "backtrack" > $O/pytest_a.log 2>&1; echo "pytest backtrack rc $?"; tail -3 $O/pytest_a.log | cut -c1-300
timeout 600 python scripts/gpu_quick.py > $O/quick.log 2>&1; echo "quick rc $?"; grep -c "^OK" $O/quick.log; grep "^FAIL" $O/quick.log | head
FUZZ_COUNT=200 timeout 900 python scripts/fuzz_parity.py 0 6701 > $O/fuzz_parity.log 2>&1; echo "fuzz parity rc $?"; grep "^FAIL" $O/fuzz_parity.log | cut -c1-300 | head -5; tail -1 $O/fuzz_parity.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "vpath2=$LQR_VPATH2 $* : "; timeout 600 python bench.py --steps 4 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for v in 1 0; do export LQR_VPATH2=$v
run --images-per-gpu 1
run --workload single4k
run --workload fhd
run --images-per-gpu 8
run --images-per-gpu 16
run --images-per-gpu 64
done
