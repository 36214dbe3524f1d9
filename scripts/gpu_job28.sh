#!/bin/bash
# k_dp_tile_p with XCD-contiguous tiles and near copies (default) against LQR_DPP_DBG=1 (no near copies) and 2 (tile = workgroup
# index, round 4's placement): parity first (the single-image suites), then single4k / fhd / config5 / 4 images
mkdir -p gpurun_out/job28; O=gpurun_out/job28
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_round3_gpu.py -m gpu -x -q > $O/parity.log 2>&1; echo "parity exit $?"; tail -2 $O/parity.log
FUZZ_COUNT=150 timeout 900 python scripts/fuzz_parity.py 0 61 > $O/fuzz_parity.log 2>&1; echo "fuzz_parity exit $?"; tail -1 $O/fuzz_parity.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "DPP_DBG=${LQR_DPP_DBG:-0} $* : "; timeout 600 python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for wl in single4k fhd config5; do for d in 0 1 2 0 2; do LQR_DPP_DBG=$d run --workload $wl; done; done
for d in 0 2; do LQR_DPP_DBG=$d run --images-per-gpu 4; done
