#!/bin/bash
# the fuzz run under LQRHIP_POISON patterns (reads of memory nothing wrote become deterministic failures)
# usage: scripts/fuzz_poison.sh SECONDS SEED PATTERN...
S=${1:-60}; SEED=${2:-778}; shift 2
mkdir -p gpurun_out/fuzz
for P in ${@:-255 0 127 128}; do
  echo "== poison $P"
  LQRHIP_POISON=$P python scripts/fuzz_parity.py $S $SEED > gpurun_out/fuzz/poison_${P}_$SEED.log 2>&1
  tail -5 gpurun_out/fuzz/poison_${P}_$SEED.log | cut -c1-260
done
