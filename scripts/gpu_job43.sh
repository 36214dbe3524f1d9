#!/bin/bash
# k_band_levels' rigidity instantiations with lean predicates (LEAN): parity, then against the last commit (liblqr-hip-ref.so)
mkdir -p gpurun_out/job43; O=gpurun_out/job43
timeout 600 python scripts/gpu_levels_quick.py > $O/quick.log 2>&1; echo "quick exit $?"; tail -1 $O/quick.log
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round3_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests exit $?"; grep -E "passed|failed" $O/tests.log | tail -1
FUZZ_COUNT=300 FUZZ_LEVELS=1 timeout 1200 python scripts/fuzz_tiles.py 0 101 > $O/fuzz_tiles_levels.log 2>&1; echo "fuzz_tiles exit $?"; tail -1 $O/fuzz_tiles_levels.log
FUZZ_COUNT=80 timeout 900 python scripts/fuzz_parity.py 0 102 0 general > $O/fuzz_parity_general.log 2>&1; echo "fuzz_parity general exit $?"; tail -1 $O/fuzz_parity_general.log
FUZZ_COUNT=30 timeout 1200 python scripts/fuzz_batch.py 0 103 > $O/fuzz_batch.log 2>&1; echo "fuzz_batch exit $?"; tail -1 $O/fuzz_batch.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
D=$PWD/gimp-lqr-plugin_amd
run() { echo -n "lib=$V $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for r in 1 2; do
  V=tree run --images-per-gpu 16 --rigidity 4; V=ref LQR_HIP_LIB=$D/liblqr-hip-ref.so run --images-per-gpu 16 --rigidity 4
  V=tree run --images-per-gpu 8; V=ref LQR_HIP_LIB=$D/liblqr-hip-ref.so run --images-per-gpu 8
done
V=tree run --images-per-gpu 16 --delta 2 --rigidity 4; V=ref LQR_HIP_LIB=$D/liblqr-hip-ref.so run --images-per-gpu 16 --delta 2 --rigidity 4
