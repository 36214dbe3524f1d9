#!/bin/bash
# k_band_levels with the six cross-lane fetches of the active-set derivation issued together (the tree) against the build two commits back (liblqr-hip-ref.so)
mkdir -p gpurun_out/job49; O=gpurun_out/job49
timeout 600 python scripts/gpu_levels_quick.py > $O/quick.log 2>&1; echo "quick exit $?"; tail -1 $O/quick.log
timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests exit $?"; tail -1 $O/tests.log
FUZZ_COUNT=300 FUZZ_LEVELS=1 timeout 1200 python scripts/fuzz_tiles.py 0 121 > $O/fuzz_tiles_levels.log 2>&1; echo "fuzz_tiles exit $?"; tail -1 $O/fuzz_tiles_levels.log
FUZZ_COUNT=60 timeout 1200 python scripts/fuzz_batch.py 0 122 > $O/fuzz_batch.log 2>&1; echo "fuzz_batch exit $?"; tail -1 $O/fuzz_batch.log
D=$PWD/gimp-lqr-plugin_amd
run() { echo -n "lib=$V $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], {k: round(v['ms']/v['launches']*1000) for k,v in d['kernels_ms'].items()})"; }
for n in 8 16; do for r in 1 2; do
  V=tree run --images-per-gpu $n
  V=ref LQR_HIP_LIB=$D/liblqr-hip-ref.so run --images-per-gpu $n
done; done
