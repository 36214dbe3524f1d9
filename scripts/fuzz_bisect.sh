#!/bin/bash
# which allocation's stale content does case $2 of seed $1 depend on?  (LQRHIP_POISON=$3, default r3)
SEED=$1; IDX=$2; P=${3:-r3}
mkdir -p gpurun_out/fuzz
run() { FUZZ_ONLY=$IDX LQRHIP_POISON=$P "$@" python scripts/fuzz_parity.py 999 $SEED 2>&1; }
LQRHIP_POISON_LOG=1 run env > gpurun_out/fuzz/bisect_log.txt
grep -c '^alloc' gpurun_out/fuzz/bisect_log.txt; grep -E '^(FAIL|ok)' gpurun_out/fuzz/bisect_log.txt | cut -c1-120
# the oracle process allocates nothing on the device: all numbers are the engine's
N=$(grep -c '^alloc' gpurun_out/fuzz/bisect_log.txt)
lo=0; hi=$N
while [ $((hi - lo)) -gt 1 ]; do
  mid=$(((lo + hi) / 2))
  if run env LQRHIP_POISON_RANGE=$lo:$mid | grep -q '^FAIL'; then hi=$mid; else lo=$mid; fi
  echo "range now $lo:$hi"
done
echo "culprit allocation: $lo"; grep "^alloc $lo " gpurun_out/fuzz/bisect_log.txt
run env LQRHIP_POISON_RANGE=$lo:$hi | grep -E '^(FAIL|ok)' | cut -c1-160
