#!/bin/bash
# A/B of engine builds (variants/*.so) on the default workload, alternating, same box
pp='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["parity_vs_oracle"] if "parity_vs_oracle" in d else None, {k:(round(v["ms"],1),v["launches"]) for k,v in d["kernels_ms"].items()})'
cp gimp-lqr-plugin_amd/liblqr-hip.so /tmp/orig.so
for rep in 1 2; do for v in "$@"; do
  echo "== $v"; cp variants/$v gimp-lqr-plugin_amd/liblqr-hip.so
  timeout -s KILL 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-times ${EXTRA} 2>&1 | tail -1 | python -c "$pp"
done; done
cp /tmp/orig.so gimp-lqr-plugin_amd/liblqr-hip.so
