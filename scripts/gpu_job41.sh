#!/bin/bash
# k_band_levels with the per-lane predicates recomputed (no scalar-register pairs held across the loop) and scalar plane bases:
# tree = that + no LDS flag reads between the barriers, v2 = that alone, ref = the last commit.  Parity of the tree first.
mkdir -p gpurun_out/job41; O=gpurun_out/job41
timeout 600 python scripts/gpu_levels_quick.py > $O/quick.log 2>&1; echo "quick exit $?"; tail -1 $O/quick.log
timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests exit $?"; tail -1 $O/tests.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
D=$PWD/gimp-lqr-plugin_amd
run() { echo -n "lib=$V $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in 8 16 64; do for r in 1 2; do
  V=tree run --images-per-gpu $n
  V=v2 LQR_HIP_LIB=$D/liblqr-hip-v2.so run --images-per-gpu $n
  V=ref LQR_HIP_LIB=$D/liblqr-hip-ref.so run --images-per-gpu $n
done; done
V=tree run --images-per-gpu 16 --rigidity 4; V=ref LQR_HIP_LIB=$D/liblqr-hip-ref.so run --images-per-gpu 16 --rigidity 4
