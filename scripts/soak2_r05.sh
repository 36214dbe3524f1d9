#!/bin/bash
# Round 5, second soak on the round's final kernels: the whole GPU suite once, then LOOPS x the three files that run child processes
# (in suite order, right after it), then the seed-7700 batch run REPS x on poisoned blocks next to a second process that loads the GPU.
# usage: scripts/soak2_r05.sh [LOOPS] [REPS]      logs: gpurun_out/soak2/
LOOPS=${1:-4}; REPS=${2:-20}
mkdir -p gpurun_out/soak2
L=gpurun_out/soak2
date > $L/summary.txt
echo "== A: full suite" | tee -a $L/summary.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > $L/A_suite.log 2>&1; echo "rc $? $(grep -E 'passed|failed' $L/A_suite.log | tail -1)" | tee -a $L/summary.txt
echo "== B: $LOOPS loops of the child-process files, in suite order" | tee -a $L/summary.txt
for i in $(seq 1 $LOOPS); do
  python -m pytest tests/test_batch_fuzz_gpu.py tests/test_recycled_blocks_gpu.py tests/test_round4_gpu.py -m gpu -q -p no:cacheprovider > $L/B_loop$i.log 2>&1
  echo "loop $i rc $? $(grep -E 'passed|failed' $L/B_loop$i.log | tail -1)" | tee -a $L/summary.txt
done
echo "== C: fuzz_batch seed 7700, poisoned blocks, next to a loading process, $REPS x (160 cases each)" | tee -a $L/summary.txt
python scripts/gpu_load.py 100000 > $L/C_load.log 2>&1 & LOADPID=$!; sleep 5
for rep in $(seq 1 $REPS); do
  q=$(( (rep % 3 == 0) ? 4 : (rep % 3 == 1) ? 8 : 16 ))
  FUZZ_COUNT=160 GPU_MAX_HW_QUEUES=$q LQRHIP_POISON=r3 python scripts/fuzz_batch.py 0 7700 > $L/C_rep$rep.log 2>&1
  echo "rep $rep queues $q rc $?: $(tail -1 $L/C_rep$rep.log)" | tee -a $L/summary.txt
  grep '^FAIL' $L/C_rep$rep.log | cut -c1-600 | tee -a $L/summary.txt
done
kill $LOADPID; wait $LOADPID 2>/dev/null
grep -h '^FAIL\|FAILED\|Error' $L/A_suite.log $L/B_loop*.log | cut -c1-400 | head -40 | tee -a $L/summary.txt
date >> $L/summary.txt
