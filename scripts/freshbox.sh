#!/bin/bash
# Both unexplained failures (round 4: test_seeded_batches[r3]; round 5: test_build_variants[-1]) happened in the first two minutes of a
# full-suite run on a FRESH box.  This is those first test files, as the first GPU work of a fresh box, then twice more on the warm box.
# usage (one gpurun call = one fresh box): gpurun -- 'bash scripts/freshbox.sh TAG'
tag=${1:-a}; mkdir -p gpurun_out/freshbox; L=gpurun_out/freshbox/$tag.log
date > $L
for i in $(seq ${PASSES:-1}); do
  python -m pytest tests/test_abi.py tests/test_batch_fuzz_gpu.py tests/test_build_variants.py -m gpu -q -p no:cacheprovider >> $L 2>&1
  echo "box $tag pass $i rc $? $(grep -E 'passed|failed' $L | tail -1)"
done
grep -E "^FAILED|three more times|FAIL\[" $L | cut -c1-600 | head -20
