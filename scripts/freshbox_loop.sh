#!/bin/bash
# N fresh boxes, one gpurun call each: the first test files (test_abi, test_batch_fuzz_gpu, test_build_variants: the second build of the
# library compiled on the box and loaded beside the first) as the FIRST GPU work of the box -- the circumstance both unexplained
# one-offs of rounds 4 / 5 shared.  Stops when gpurun_out/freshbox/STOP exists.   scripts/freshbox_loop.sh N TAGPREFIX
n=${1:-10}; pre=${2:-fb}
mkdir -p gpurun_out/freshbox
for i in $(seq $n); do
  [ -e gpurun_out/freshbox/STOP ] && break
  /usr/local/graft/bin/gpurun --timeout 600 -- "bash scripts/freshbox.sh ${pre}_$i" > /tmp/fb_${pre}_$i.log 2>&1
  grep -E "^box |status=" /tmp/fb_${pre}_$i.log | tr '\n' ' ' >> gpurun_out/freshbox/loop_${pre}.txt; echo >> gpurun_out/freshbox/loop_${pre}.txt
  grep -q "status=transient" /tmp/fb_${pre}_$i.log && sleep 120
done
