#!/bin/bash
# the fuzz run in every flavour: plain, recycled-block contents randomised (LQRHIP_POISON=r3, r1), and the general
# (delta_x 2 / rigidity mask) bend of both.  usage: scripts/fuzz_all.sh SECONDS_EACH SEED
S=${1:-90}; SEED=${2:-1}
mkdir -p gpurun_out/fuzz
rc=0
for flavour in "plain::" "r3:r3:" "r1:r1:" "general::0 general" "general_r3:r3:0 general"; do
  IFS=: read name poison extra <<< "$flavour"
  LQRHIP_POISON=$poison python scripts/fuzz_parity.py $S $SEED $extra > gpurun_out/fuzz/all_${name}_$SEED.log 2>&1 || rc=1
  echo "== $name: $(tail -1 gpurun_out/fuzz/all_${name}_$SEED.log)"
  grep '^FAIL' gpurun_out/fuzz/all_${name}_$SEED.log | cut -c1-330 | head -8
done
exit $rc
