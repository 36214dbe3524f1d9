#!/bin/bash
# default workload under a few env settings: "VAR=val,VAR2=val2" per argument
pp='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:(round(v["ms"],1),v["launches"]) for k,v in d["kernels_ms"].items()})'
for e in "$@"; do echo "== $e"; env $(echo $e | tr ',' ' ') timeout -s KILL 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-times 2>&1 | tail -1 | python -c "$pp"; done
