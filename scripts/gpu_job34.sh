#!/bin/bash
# same-box A/B: the tree against gimp-lqr-plugin_amd/liblqr-hip-ref.so (built from the last commit under variants/ref)
mkdir -p gpurun_out/job34; O=gpurun_out/job34
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
G=$PWD/gimp-lqr-plugin_amd/liblqr-hip-ref.so
run() { echo -n "lib=${LQR_HIP_LIB:+ref} $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 16 32 48; do run --images-per-gpu $n; LQR_HIP_LIB=$G run --images-per-gpu $n; run --images-per-gpu $n; LQR_HIP_LIB=$G run --images-per-gpu $n; done
