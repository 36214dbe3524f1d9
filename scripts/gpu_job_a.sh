mkdir -p gpurun_out/r06a; O=gpurun_out/r06a
timeout 900 python -m pytest tests/test_faults_gpu.py -m gpu -x -q -p no:cacheprovider > $O/faults.log 2>&1; echo "faults rc $? $(tail -1 $O/faults.log)"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_faults_gpu.py > $O/suite.log 2>&1; echo "suite rc $? $(grep -E 'passed|failed' $O/suite.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/suite.log | head -20
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); print('driver line', d['value'], d['ms_per_step'], d['summary'], d['roofline']['frac'])"
