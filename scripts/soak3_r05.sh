#!/bin/bash
# Round 5, final soak on the round's final kernels: the whole GPU suite, then count-bounded fuzz runs with their own seeds (k_band_levels is
# in every mix: a third of the batch cases, every fourth single-image case, all of the FUZZ_LEVELS tile cases), plain and on poisoned blocks.
mkdir -p gpurun_out/soak3; L=gpurun_out/soak3
date > $L/summary.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > $L/suite.log 2>&1; echo "suite rc $? $(grep -E 'passed|failed' $L/suite.log | tail -1)" | tee -a $L/summary.txt
run() { name=$1; shift; "$@" > $L/$name.log 2>&1; echo "$name rc $?: $(tail -1 $L/$name.log)" | tee -a $L/summary.txt; grep '^FAIL' $L/$name.log | cut -c1-500 | head -5 | tee -a $L/summary.txt; }
run levels_plain    env FUZZ_LEVELS=1 FUZZ_COUNT=1200 python scripts/fuzz_tiles.py 0 50501
run levels_poison   env FUZZ_LEVELS=1 FUZZ_COUNT=800 LQRHIP_POISON=r3 python scripts/fuzz_tiles.py 0 50502
run tiles_plain     env FUZZ_COUNT=400 python scripts/fuzz_tiles.py 0 50503
run batch_plain     env FUZZ_COUNT=600 GPU_MAX_HW_QUEUES=8 python scripts/fuzz_batch.py 0 50504
run batch_poison    env FUZZ_COUNT=500 GPU_MAX_HW_QUEUES=16 LQRHIP_POISON=r3 python scripts/fuzz_batch.py 0 50505
run parity_plain    env FUZZ_COUNT=700 python scripts/fuzz_parity.py 0 50506
run parity_extras   env FUZZ_COUNT=300 FUZZ_EXTRAS=1 python scripts/fuzz_parity.py 0 50507
run parity_general  env FUZZ_COUNT=400 python scripts/fuzz_parity.py 0 50508 0 general
run general_poison  env FUZZ_COUNT=300 LQRHIP_POISON=r3 python scripts/fuzz_parity.py 0 50509 0 general
run interactive     env FUZZ_COUNT=1500 python scripts/fuzz_interactive.py 0 50510
date >> $L/summary.txt
