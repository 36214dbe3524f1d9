#!/bin/bash
# an hour of the one case that failed once (soak 3), on both builds, two processes side by side (each other's load)
mkdir -p gpurun_out/job22; O=gpurun_out/job22
NS=$PWD/tests/c/build/liblqr-hip-default-sched.so
LQR_HIP_LIB=$NS python scripts/repro_buildvariant.py 1500s > $O/nosched_a.log 2>&1 &
python scripts/repro_buildvariant.py 1500s > $O/default_a.log 2>&1 &
wait
LQR_HIP_LIB=$NS python scripts/repro_buildvariant.py 1000s 420 260 380 240 > $O/nosched_b.log 2>&1 &
LQR_HIP_LIB=$NS python scripts/repro_buildvariant.py 1000s 300 160 260 140 > $O/nosched_c.log 2>&1 &
wait
for f in nosched_a default_a nosched_b nosched_c; do echo "$f: $(tail -1 $O/$f.log)"; grep -A1 MISMATCH $O/$f.log | head -8; done
