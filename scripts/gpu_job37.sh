#!/bin/bash
# k_band_update_tw with scalar plane bases (uni_ptr) against the last commit's build (liblqr-hip-ref.so), 64 and 128 images; parity first
mkdir -p gpurun_out/job37; O=gpurun_out/job37
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests exit $?"; tail -2 $O/tests.log
FUZZ_COUNT=40 timeout 1200 python scripts/fuzz_batch.py 0 91 > $O/fuzz_batch.log 2>&1; echo "fuzz_batch exit $?"; tail -1 $O/fuzz_batch.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
G=$PWD/gimp-lqr-plugin_amd/liblqr-hip-ref.so
run() { echo -n "lib=${LQR_HIP_LIB:+ref} $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for r in 1 2 3; do run --images-per-gpu 64; LQR_HIP_LIB=$G run --images-per-gpu 64; done
run --images-per-gpu 64 --update-mode 0; LQR_HIP_LIB=$G run --images-per-gpu 64 --update-mode 0
run --images-per-gpu 16 --update-mode 0; LQR_HIP_LIB=$G run --images-per-gpu 16 --update-mode 0
