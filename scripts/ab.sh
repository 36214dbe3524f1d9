#!/bin/bash
# One parameterised A/B run (replaces the per-experiment gpu_job*.sh of rounds 4-5).  Every comparison in DESIGN.md / NOTES is made
# this way: both sides on ONE box, alternating, >= 3 alternations, short bench runs with the per-kernel times on the line.
#   scripts/ab.sh TAG "A: bench args" "B: bench args" [reps=3] [images="64"]
# env: LIB_A / LIB_B = another build of the library for that side (LQR_HIP_LIB), GATE=1 = parity gate first (hand-made level cases,
# tests/test_round5_gpu.py, count-bounded fuzz) -- a variant that fails the gate is not timed; STEPS / WARMUP (default 3 / 1).
tag=$1; A=$2; B=$3; reps=${4:-3}; images=${5:-64}
O=gpurun_out/$tag; mkdir -p $O
if [ -n "$GATE" ]; then
  timeout 600 python scripts/gpu_levels_quick.py > $O/quick.log 2>&1; q=$?; echo "gate: levels quick exit $q $(tail -1 $O/quick.log)"
  timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_faults_gpu.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; t=$?; echo "gate: tests exit $t $(tail -1 $O/tests.log)"
  FUZZ_COUNT=${FUZZ_N:-300} timeout 1200 python scripts/fuzz_levels.py 0 ${SEED:-131} > $O/fuzz_levels.log 2>&1; f=$?; echo "gate: fuzz_levels exit $f $(tail -1 $O/fuzz_levels.log)"
  FUZZ_COUNT=60 timeout 1200 python scripts/fuzz_batch.py 0 $((${SEED:-131} + 1)) > $O/fuzz_batch.log 2>&1; g=$?; echo "gate: fuzz_batch exit $g $(tail -1 $O/fuzz_batch.log)"
  [ $q -ne 0 -o $t -ne 0 -o $f -ne 0 -o $g -ne 0 ] && { echo "gate FAILED: not timed"; exit 1; }
fi
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d.get("kernels_ms", {}).items()})'
run() { side=$1; lib=$2; shift; shift; echo -n "$side [$*] : "; LQR_HIP_LIB=$lib timeout 600 python bench.py --steps ${STEPS:-3} --warmup ${WARMUP:-1} --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for n in $images; do for r in $(seq $reps); do
  run A "$LIB_A" --images-per-gpu $n $A
  run B "$LIB_B" --images-per-gpu $n $B
done; done | tee $O/ab.txt
