#!/bin/bash
mkdir -p gpurun_out/job11; O=gpurun_out/job11
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_round2_gpu.py tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_a.log
FUZZ_COUNT=300 timeout 900 python scripts/fuzz_parity.py 0 6601 > $O/fuzz_parity.log 2>&1; echo "fuzz parity rc $?"; grep "^FAIL" $O/fuzz_parity.log | cut -c1-400 | head; tail -1 $O/fuzz_parity.log
FUZZ_COUNT=200 timeout 900 python scripts/fuzz_parity.py 0 6602 0 general > $O/fuzz_general.log 2>&1; echo "fuzz general rc $?"; grep "^FAIL" $O/fuzz_general.log | cut -c1-400 | head; tail -1 $O/fuzz_general.log
FUZZ_COUNT=160 GPU_MAX_HW_QUEUES=8 timeout 900 python scripts/fuzz_batch.py 0 6603 > $O/fuzz_batch.log 2>&1; echo "fuzz batch rc $?"; grep "^FAIL" $O/fuzz_batch.log | cut -c1-400 | head; tail -1 $O/fuzz_batch.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()}, d.get("band_levels_stats"))'
run() { echo -n "$* : "; timeout 600 python bench.py --steps 2 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
run --images-per-gpu 64
run --images-per-gpu 16 --delta 2
run --images-per-gpu 16 --delta 3
run --images-per-gpu 16 --delta 2 --rigidity 5
run --images-per-gpu 8 --delta 2
tail -3 $O/bench.err
