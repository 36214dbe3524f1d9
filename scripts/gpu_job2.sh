#!/bin/bash
# round 5, job 2: the split build runs (cross-unit kernel launches), k_band_levels parity + first timings
mkdir -p gpurun_out/job2; O=gpurun_out/job2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout 600 python scripts/gpu_levels_quick.py > $O/levels_quick.log 2>&1; echo "levels quick rc $?"; grep -c "^ok" $O/levels_quick.log; grep "^FAIL" $O/levels_quick.log | head -20; tail -1 $O/levels_quick.log
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_tolerance_boundary.py tests/test_round4_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "pytest a rc $?"; tail -3 $O/pytest_a.log
FUZZ_LEVELS=1 FUZZ_COUNT=300 timeout 900 python scripts/fuzz_tiles.py 0 5500 > $O/fuzz_levels.log 2>&1; echo "fuzz levels rc $?"; grep "^FAIL" $O/fuzz_levels.log | cut -c1-400 | head -10; tail -1 $O/fuzz_levels.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()}, d.get("band_levels_stats"), d.get("band_tiles_stats"))'
run() { echo -n "$* : "; timeout 300 python bench.py --steps 4 --warmup 2 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
run --images-per-gpu 64
run --images-per-gpu 64 --update-mode 5 --band-levels 6
run --images-per-gpu 64 --update-mode 5 --band-levels 8
run --images-per-gpu 64 --update-mode 5 --band-levels 4
run --images-per-gpu 16
run --images-per-gpu 16 --update-mode 5 --band-levels 8
run --images-per-gpu 16 --update-mode 5 --band-levels 12
run --images-per-gpu 8
run --images-per-gpu 8 --update-mode 5 --band-levels 12
run --images-per-gpu 8 --update-mode 5 --band-levels 16
run --images-per-gpu 1 --update-mode 5 --band-levels 16
run --images-per-gpu 1
