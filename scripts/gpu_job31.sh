#!/bin/bash
# LDS flags as ds_read / ds_write (default build) against the old FLAT form (liblqr-hip-genflag.so, make EXTRA=-DLDS_FLAG_GENERIC):
# parity of the default build first, then alternating runs
mkdir -p gpurun_out/job31; O=gpurun_out/job31
timeout 600 python scripts/gpu_levels_quick.py > $O/quick.log 2>&1; echo "quick exit $?"; tail -1 $O/quick.log
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round3_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests exit $?"; tail -2 $O/tests.log
FUZZ_COUNT=200 FUZZ_LEVELS=1 timeout 1200 python scripts/fuzz_tiles.py 0 71 > $O/fuzz_tiles_levels.log 2>&1; echo "fuzz_tiles exit $?"; tail -1 $O/fuzz_tiles_levels.log
FUZZ_COUNT=100 timeout 900 python scripts/fuzz_parity.py 0 72 > $O/fuzz_parity.log 2>&1; echo "fuzz_parity exit $?"; tail -1 $O/fuzz_parity.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
G=$PWD/gimp-lqr-plugin_amd/liblqr-hip-genflag.so
run() { echo -n "lib=${LQR_HIP_LIB:+genflag} $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 16 48; do run --images-per-gpu $n; LQR_HIP_LIB=$G run --images-per-gpu $n; run --images-per-gpu $n; LQR_HIP_LIB=$G run --images-per-gpu $n; done
for wl in single4k config5; do run --workload $wl; LQR_HIP_LIB=$G run --workload $wl; run --workload $wl; LQR_HIP_LIB=$G run --workload $wl; done
run --images-per-gpu 4; LQR_HIP_LIB=$G run --images-per-gpu 4
