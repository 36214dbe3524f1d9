#!/bin/bash
# N fresh boxes, one gpurun call each, the WHOLE -m gpu suite as the box's first GPU work (the circumstance of both unexplained one-offs:
# a full-suite run on a fresh box), with the harness' mismatch capture armed (tests/harness.py save_mismatch -> gpurun_out/mismatch/).
#   scripts/freshsuite_loop.sh N TAGPREFIX      stops when gpurun_out/freshsuite/STOP exists
n=${1:-4}; pre=${2:-fs}
mkdir -p gpurun_out/freshsuite
for i in $(seq $n); do
  [ -e gpurun_out/freshsuite/STOP ] && break
  /usr/local/graft/bin/gpurun --timeout 1200 -- "mkdir -p gpurun_out/freshsuite; date > gpurun_out/freshsuite/${pre}_$i.log; python -m pytest tests -m gpu -q -p no:cacheprovider >> gpurun_out/freshsuite/${pre}_$i.log 2>&1; echo \"box ${pre}_$i rc \$? \$(grep -E 'passed|failed' gpurun_out/freshsuite/${pre}_$i.log | tail -1)\"; grep -E '^FAILED|three more times' gpurun_out/freshsuite/${pre}_$i.log | cut -c1-400 | head -5" > /tmp/fs_${pre}_$i.log 2>&1
  grep -E "^box |status=|^FAILED" /tmp/fs_${pre}_$i.log | tr '\n' ' ' >> gpurun_out/freshsuite/loop_${pre}.txt; echo >> gpurun_out/freshsuite/loop_${pre}.txt
  grep -q "status=transient" /tmp/fs_${pre}_$i.log && sleep 120
done
