#!/bin/bash
# Round 5, VERDICT r4 item 1: run down the one unexplained failure of tests/test_batch_fuzz_gpu.py::test_seeded_batches[r3].
# Everything is count-bounded now (tests/fuzz_common.py) and every child's whole output is kept.
#   A  the whole GPU suite once, in order (the condition the failure happened under)
#   B  LOOPS x the three files that run child processes, in suite order, right after it
#   C  the failing run itself (fuzz_batch seed 7700, 160 cases; plain and on poisoned blocks) with 4 / 8 / 16 hardware queues,
#      alone and next to a second process that loads the GPU (scripts/gpu_load.py)
# usage: scripts/soak_r05.sh [LOOPS] [REPS]      logs: gpurun_out/soak/
LOOPS=${1:-3}; REPS=${2:-1}
mkdir -p gpurun_out/soak
L=gpurun_out/soak
date > $L/summary.txt
echo "== A: full suite" | tee -a $L/summary.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > $L/A_suite.log 2>&1; echo "rc $? $(tail -1 $L/A_suite.log)" | tee -a $L/summary.txt
echo "== B: $LOOPS loops of the child-process files, in suite order" | tee -a $L/summary.txt
for i in $(seq 1 $LOOPS); do
  python -m pytest tests/test_batch_fuzz_gpu.py tests/test_recycled_blocks_gpu.py tests/test_round4_gpu.py -m gpu -q -p no:cacheprovider > $L/B_loop$i.log 2>&1
  echo "loop $i rc $? $(tail -1 $L/B_loop$i.log)" | tee -a $L/summary.txt
done
echo "== C: fuzz_batch seed 7700 x queues x load" | tee -a $L/summary.txt
for rep in $(seq 1 $REPS); do
for load in 0 1; do
  if [ $load = 1 ]; then python scripts/gpu_load.py 100000 > $L/C_load.log 2>&1 & LOADPID=$!; sleep 5; fi
  for q in 4 8 16; do for poison in "" r3; do
    t0=$(date +%s)
    FUZZ_COUNT=160 GPU_MAX_HW_QUEUES=$q LQRHIP_POISON=$poison python scripts/fuzz_batch.py 0 7700 > $L/C_q${q}_p${poison:-none}_load${load}_rep$rep.log 2>&1
    echo "queues $q poison '${poison}' load $load rep $rep rc $? $(( $(date +%s) - t0 )) s: $(tail -1 $L/C_q${q}_p${poison:-none}_load${load}_rep$rep.log)" | tee -a $L/summary.txt
    grep '^FAIL' $L/C_q${q}_p${poison:-none}_load${load}_rep$rep.log | cut -c1-600 | tee -a $L/summary.txt
  done; done
  if [ $load = 1 ]; then kill $LOADPID; wait $LOADPID 2>/dev/null; fi
done
done
grep -h '^FAIL\|FAILED\|Error' $L/A_suite.log $L/B_loop*.log | cut -c1-400 | head -40 | tee -a $L/summary.txt
date >> $L/summary.txt
