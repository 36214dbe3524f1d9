#!/bin/bash
# The soak: the whole GPU suite, then count-bounded fuzz runs with their own seeds, plain, on poisoned blocks, and with an
# image's slots deliberately on different XCDs (LQR_LV_DBG=8: the near copies are written and never seen).
mkdir -p gpurun_out/soak; L=gpurun_out/soak
date > $L/summary.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > $L/suite.log 2>&1; echo "suite rc $? $(grep -E 'passed|failed' $L/suite.log | tail -1)" | tee -a $L/summary.txt
run() { name=$1; shift; "$@" > $L/$name.log 2>&1; echo "$name rc $?: $(tail -1 $L/$name.log)" | tee -a $L/summary.txt; grep '^FAIL' $L/$name.log | cut -c1-500 | head -5 | tee -a $L/summary.txt; }
run levels_plain    env FUZZ_COUNT=900 python scripts/fuzz_levels.py 0 60601
run levels_poison   env FUZZ_COUNT=500 LQRHIP_POISON=r3 python scripts/fuzz_levels.py 0 60602
run levels_crossxcd env FUZZ_COUNT=500 LQR_LV_DBG=8 python scripts/fuzz_levels.py 0 60603
run batch_plain     env FUZZ_COUNT=400 GPU_MAX_HW_QUEUES=8 python scripts/fuzz_batch.py 0 60604
run batch_poison    env FUZZ_COUNT=300 GPU_MAX_HW_QUEUES=16 LQRHIP_POISON=r3 python scripts/fuzz_batch.py 0 60605
run batch_crossxcd  env FUZZ_COUNT=200 LQR_LV_DBG=8 python scripts/fuzz_batch.py 0 60606
run parity_plain    env FUZZ_COUNT=600 python scripts/fuzz_parity.py 0 60607
run parity_general  env FUZZ_COUNT=400 python scripts/fuzz_parity.py 0 60608 0 general
run general_poison  env FUZZ_COUNT=200 LQRHIP_POISON=r3 python scripts/fuzz_parity.py 0 60609 0 general
run interactive     env FUZZ_COUNT=1000 python scripts/fuzz_interactive.py 0 60610
run parity_vp       env FUZZ_COUNT=400 LQR_VP=1 python scripts/fuzz_parity.py 0 60611             # the parallel backtrack forced for every case
run parity_extras   env FUZZ_COUNT=400 FUZZ_EXTRAS=1 python scripts/fuzz_parity.py 0 60612        # the plug-in's other switches
date >> $L/summary.txt
