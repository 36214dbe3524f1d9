#!/bin/bash
mkdir -p gpurun_out/job15; O=gpurun_out/job15
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_a.log | tail -1; grep -E "^FAILED|Error" $O/pytest_a.log | head -5
FUZZ_COUNT=150 timeout 900 python scripts/fuzz_parity.py 0 6801 0 general > $O/fuzz_general.log 2>&1; echo "fuzz general rc $?"; tail -1 $O/fuzz_general.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "$* : "; timeout 600 python bench.py --steps 2 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
run --images-per-gpu 16 --delta 2
run --images-per-gpu 16 --delta 2 --rigidity 5
run --images-per-gpu 16 --delta 3
run --images-per-gpu 8 --delta 2
