#!/bin/bash
# the round's last GPU run: the whole suite on the final tree, the driver's exact command, the group sizes whose schedule changed last
mkdir -p gpurun_out/final; O=gpurun_out/final
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/suite.log 2>&1; echo "suite rc $? $(grep -E 'passed|failed' $O/suite.log | tail -1)"; grep -E "^FAILED|three more times" $O/suite.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2> $O/driver_cmd_bench.err; python3 -c "
import json; d=json.load(open('$O/driver_cmd_bench.json')); print('driver line', d['value'], d['ms_per_step'], d['summary'], d['roofline']['frac'])"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], d["config"]["streams_per_gpu"])'
for n in 24 32 40 48; do echo -n "$n images: "; timeout 300 python bench.py --steps 5 --warmup 2 --images-per-gpu $n --no-configs --no-cpu-baseline --no-phases 2>/dev/null | python3 -c "$P"; done
