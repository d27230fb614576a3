#!/bin/bash
# round 4, GPU call 33: the final tree (stem2 with the rotated depthwise-1 map as default): whole suite, smoke(), the default bench line, the driver's invocation, kbench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c33
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time timeout 600 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err ) 2> $O/bench_time.txt; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 200 python tools/kbench.py --n 256 --tag r4c33_fp16 > $O/kbench_fp16.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-pmc > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/e5.err
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; tail -1 $O/smoke.log; tail -3 $O/bench_time.txt
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
grep -h "==" $O/kbench_*.txt
