#!/bin/bash
cp gimp-lqr-plugin_amd/liblqr-hip.so /tmp/orig.so
cp variants/liblqr-hip-tws.so gimp-lqr-plugin_amd/liblqr-hip.so
LQRHIP_TILED_UPDATE_PX=0 timeout -s KILL 120 python bench.py --workload fhd --steps 1 --warmup 0 --no-cpu-baseline --kernel-times 2>&1 | grep "tw wave" | sed -n '1,6p;100,104p;390,396p'
LQRHIP_TILED_UPDATE_PX=0 timeout -s KILL 120 python bench.py --images-per-gpu 1 --steps 1 --warmup 0 --no-cpu-baseline --kernel-times 2>&1 | grep "tw wave" | sed -n '1,4p;100,104p;390,396p'
cp /tmp/orig.so gimp-lqr-plugin_amd/liblqr-hip.so
