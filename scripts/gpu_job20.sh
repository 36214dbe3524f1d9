#!/bin/bash
mkdir -p gpurun_out/job20; O=gpurun_out/job20
make -C gimp-lqr-plugin_amd -j8 SCHED= BUILD=$PWD/tests/c/build/nosched OUT=$PWD/tests/c/build/liblqr-hip-default-sched.so > $O/make.log 2>&1; echo "make rc $?"
LQR_HIP_LIB=$PWD/tests/c/build/liblqr-hip-default-sched.so timeout 900 python scripts/repro_buildvariant.py 400 > $O/nosched.log 2>&1; tail -3 $O/nosched.log; grep -c MISMATCH $O/nosched.log
timeout 900 python scripts/repro_buildvariant.py 400 > $O/default.log 2>&1; tail -2 $O/default.log; grep -c MISMATCH $O/default.log
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_build_variants.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1; done
