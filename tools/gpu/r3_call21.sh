#!/bin/bash
# round 3, GPU call 21: the final tree once more -- whole suite, smoke(), the configs[4]-shape int8 line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c21
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 200 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
grep -v "compute time" $O/pytest.log | grep -E "passed|failed" | tail -2; tail -1 $O/smoke.log
python -c "
import json; j=json.load(open('$O/bench_int8_mnet25_b32.json')); print('int8 mnet25 b32', round(j['images_per_sec']), round(j['value']))"
