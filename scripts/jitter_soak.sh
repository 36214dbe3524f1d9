#!/bin/bash
# Round 6: the hunt for the two unexplained one-offs of rounds 4 / 5 with the interleavings forced instead of waited for.
# liblqr-hip-jitter.so (make EXTRA=-DLQR_JITTER BUILD=build_jitter OUT=liblqr-hip-jitter.so) sleeps a pseudo-random 0 .. ~120 k cycles at
# every hand-over site of the two spin protocols (k_dp_tile_p: poll, publish, barrier, deferred store; k_band_levels: level barrier,
# store, prefetch, granules, word) -- thousands of unusual interleavings per launch.  Every result against the oracle.
#   scripts/jitter_soak.sh [TAG] [scale]
tag=${1:-jitter}; s=${2:-1}
O=gpurun_out/$tag; mkdir -p $O
export LQR_HIP_LIB=$PWD/gimp-lqr-plugin_amd/liblqr-hip-jitter.so LQR_DPP_DBG=4 LQR_LV_DBG=16
run() { name=$1; shift; "$@" > $O/$name.log 2>&1; echo "$name rc $?: $(tail -1 $O/$name.log)" | tee -a $O/summary.txt; grep -E '^FAIL|MISMATCH' $O/$name.log | cut -c1-700 | head -5 | tee -a $O/summary.txt; }
date > $O/summary.txt
run case_r5        python scripts/repro_buildvariant.py $((60 * s))s
run case_r5_b      python scripts/repro_buildvariant.py $((40 * s))s 520 300 470 270
run parity         env FUZZ_COUNT=$((300 * s)) python scripts/fuzz_parity.py 0 70701
run parity_general env FUZZ_COUNT=$((150 * s)) python scripts/fuzz_parity.py 0 70702 0 general
run levels         env FUZZ_COUNT=$((300 * s)) python scripts/fuzz_levels.py 0 70703
run batch          env FUZZ_COUNT=$((80 * s)) python scripts/fuzz_batch.py 0 70704
run batch_poison   env FUZZ_COUNT=$((60 * s)) LQRHIP_POISON=r3 python scripts/fuzz_batch.py 0 70705
date >> $O/summary.txt
