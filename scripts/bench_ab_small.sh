#!/bin/bash
# A/B of engine builds (variants/*.so) on the single-4K workload, same box
pp='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:(round(v["ms"],1),v["launches"]) for k,v in d["kernels_ms"].items()})'
cp gimp-lqr-plugin_amd/liblqr-hip.so /tmp/orig.so
for v in "$@"; do
  echo "== $v"; cp variants/$v gimp-lqr-plugin_amd/liblqr-hip.so
  timeout -s KILL 120 python bench.py --workload single4k --steps 2 --warmup 1 --no-cpu-baseline --kernel-times 2>&1 | tail -1 | python -c "$pp"
  timeout -s KILL 120 python bench.py --images-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline --kernel-times 2>&1 | tail -1 | python -c "$pp"
done
cp /tmp/orig.so gimp-lqr-plugin_amd/liblqr-hip.so
