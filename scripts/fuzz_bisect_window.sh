#!/bin/bash
# which window of allocation $3 (a plane of stride $4, $5 rows) does case $2 of seed $1 depend on?
SEED=$1; IDX=$2; A=$3; ST=$4; ROWS=$5
run() { FUZZ_ONLY=$IDX LQRHIP_POISON=r3 LQRHIP_POISON_RANGE=$A:$((A+1)) LQRHIP_POISON_WINDOW=$1 python scripts/fuzz_parity.py 999 $SEED 2>&1 | grep -q '^FAIL'; }
run $ST:0:$ST:0:$ROWS && echo "whole window fails" || { echo "whole window passes?"; exit 1; }
lo=0; hi=$ST          # smallest c1 that still fails with columns [0, c1)
while [ $((hi - lo)) -gt 1 ]; do mid=$(((lo + hi) / 2)); if run $ST:0:$mid:0:$ROWS; then hi=$mid; else lo=$mid; fi; done
C1=$hi; echo "columns [0,$C1) needed"
lo=0; hi=$C1          # largest c0 that still fails with [c0, C1)
while [ $((hi - lo)) -gt 1 ]; do mid=$(((lo + hi) / 2)); if run $ST:$mid:$C1:0:$ROWS; then lo=$mid; else hi=$mid; fi; done
C0=$lo; echo "columns [$C0,$C1)"
lo=0; hi=$ROWS
while [ $((hi - lo)) -gt 1 ]; do mid=$(((lo + hi) / 2)); if run $ST:$C0:$C1:0:$mid; then hi=$mid; else lo=$mid; fi; done
R1=$hi
lo=0; hi=$R1
while [ $((hi - lo)) -gt 1 ]; do mid=$(((lo + hi) / 2)); if run $ST:$C0:$C1:$mid:$R1; then lo=$mid; else hi=$mid; fi; done
echo "window: columns [$C0,$C1) rows [$lo,$R1)"
