#!/bin/bash
# the in-suite condition of the one failure of tests/test_build_variants.py: the shipped library initialised in the same process (tests/test_abi.py
# does that) and the second build compiled on the box right before
mkdir -p gpurun_out/job23; O=gpurun_out/job23
for i in 1 2 3 4 5 6; do
  rm -rf tests/c/build/nosched tests/c/build/liblqr-hip-default-sched.so
  python -m pytest tests/test_abi.py tests/test_build_variants.py -m gpu -q -p no:cacheprovider > $O/rebuild_$i.log 2>&1
  echo "rebuilt on the box, pass $i: rc $? $(grep -E 'passed|failed' $O/rebuild_$i.log | tail -1)"; grep -E "three more times|^E   " $O/rebuild_$i.log | head -4
done
for i in $(seq 1 25); do
  python -m pytest tests/test_abi.py tests/test_build_variants.py -m gpu -q -p no:cacheprovider > $O/warm_$i.log 2>&1 || { echo "warm pass $i FAILED"; grep -E "three more times|^E   " $O/warm_$i.log | head -4; }
done
echo "25 passes without rebuilding done: $(grep -l ' passed' $O/warm_*.log | wc -l) logs with 'passed', $(grep -l 'failed' $O/warm_*.log | wc -l) with 'failed'"
