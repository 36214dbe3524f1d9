#!/bin/bash
# Shader clock and power while the bench runs: rocm-smi sampled every 0.25 s next to bench.py (64 x 4K on 4 streams, then 16 x 4K, then one 4K image)
mkdir -p gpurun_out/clocks; O=gpurun_out/clocks
sample() { while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use|fclk" | tr -s ' ' | tr '\n' '|'; echo; sleep 0.25; done; }
for cfg in "--images-per-gpu 64 --steps 30 --warmup 3" "--images-per-gpu 16 --steps 40 --warmup 3" "--workload single4k --steps 6 --warmup 1" "--images-per-gpu 64 --sub-batches 1 --steps 20 --warmup 3"; do
  name=$(echo $cfg | tr -d ' -' | cut -c1-28)
  sample > $O/smi_$name.log & SP=$!
  sleep 1
  python bench.py $cfg --no-configs --no-cpu-baseline --no-phases --no-kernel-breakdown 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])"
  kill $SP; wait $SP 2>/dev/null
  echo "== $name"; sort $O/smi_$name.log | uniq -c | sort -rn | head -6 | cut -c1-400
done
