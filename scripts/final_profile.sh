#!/bin/bash
# bench line + rocprofv3 kernel stats of the same command, for profiles/
tag=${1:-r01d}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
timeout -s KILL 400 python $R/bench.py > $R/gpurun_out/$tag/bench.json 2> $R/gpurun_out/$tag/bench.err
tail -1 $R/gpurun_out/$tag/bench.json | cut -c1-2500
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/$tag/bench_prof.log 2>&1
find $R/gpurun_out/$tag/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
head -12 $R/gpurun_out/$tag/kernel_stats.csv
