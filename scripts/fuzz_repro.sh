#!/bin/bash
# fuzz run, then every failing case again (3 times in every update mode) to tell a deterministic defect from a race
# usage: scripts/fuzz_repro.sh SECONDS SEED [general]
S=${1:-120}; SEED=${2:-1}; G=${3:+0 general}
mkdir -p gpurun_out/fuzz
python scripts/fuzz_parity.py $S $SEED $G > gpurun_out/fuzz/run_$SEED.log 2>&1
tail -4 gpurun_out/fuzz/run_$SEED.log | cut -c1-300
IDX=$(grep '^FAIL case' gpurun_out/fuzz/run_$SEED.log | awk '{print $3}' | sort -un | head -8 | paste -sd,)
[ -z "$IDX" ] && exit 0
echo "failing indices: $IDX"
FUZZ_ONLY=$IDX FUZZ_REPEAT=3 FUZZ_MODES=-1,0,1,2,3 python scripts/fuzz_parity.py 9999 $SEED $G 2>&1 | cut -c1-260 | tee gpurun_out/fuzz/repro_$SEED.log | tail -80
