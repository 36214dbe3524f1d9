#!/bin/bash
mkdir -p gpurun_out/job21; O=gpurun_out/job21
NS=$PWD/tests/c/build/liblqr-hip-default-sched.so
# 1. the new backtrack kernel
timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -q -x -p no:cacheprovider -k "backtrack" > $O/pytest_bt.log 2>&1; echo "pytest backtrack rc $?"; grep -E "passed|failed" $O/pytest_bt.log | tail -1; grep -E "^E  " $O/pytest_bt.log | head -5
FUZZ_COUNT=250 timeout 900 python scripts/fuzz_parity.py 0 6901 > $O/fuzz_parity.log 2>&1; echo "fuzz parity rc $?"; grep "^FAIL" $O/fuzz_parity.log | cut -c1-300 | head -5; tail -1 $O/fuzz_parity.log
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], {k: round(v["ms"]/v["launches"]*1000) for k,v in d["kernels_ms"].items()})'
run() { echo -n "vpath2=$LQR_VPATH2 $* : "; timeout 600 python bench.py --steps 4 --warmup 1 --no-configs --no-cpu-baseline --no-phases --kernel-times "$@" 2>>$O/bench.err | python3 -c "$P"; }
for v in 1 0; do export LQR_VPATH2=$v
run --images-per-gpu 1
run --workload fhd
run --images-per-gpu 8
run --images-per-gpu 16
run --images-per-gpu 64
done
unset LQR_VPATH2
# 2. the default-scheduler build: seeded cases on poisoned and plain blocks, then cold starts on dirty device memory
LQR_HIP_LIB=$NS LQRHIP_POISON=r3 FUZZ_COUNT=600 timeout 1200 python scripts/fuzz_parity.py 0 6902 > $O/ns_poison.log 2>&1; echo "nosched poisoned rc $?"; grep "^FAIL" $O/ns_poison.log | cut -c1-400 | head -5; tail -1 $O/ns_poison.log
LQR_HIP_LIB=$NS FUZZ_COUNT=600 timeout 1200 python scripts/fuzz_parity.py 0 6903 > $O/ns_plain.log 2>&1; echo "nosched plain rc $?"; grep "^FAIL" $O/ns_plain.log | cut -c1-400 | head -5; tail -1 $O/ns_plain.log
bad=0
for i in $(seq 1 40); do
  LQRHIP_POISON=r3 FUZZ_COUNT=4 python scripts/fuzz_parity.py 0 $((7000+i)) > /dev/null 2>&1
  LQR_HIP_LIB=$NS python scripts/repro_buildvariant.py 2 > $O/cold_$i.log 2>&1
  if ! grep -q " 0 mismatches" $O/cold_$i.log; then bad=$((bad+1)); echo "cold start $i:"; grep MISMATCH $O/cold_$i.log | head -3; fi
done
echo "cold starts on dirty memory: 40 processes, $bad with mismatches"
