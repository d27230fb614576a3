#!/bin/bash
# round 3, GPU call 19: final tree -- whole suite, smoke(), the int8 bench lines and the default line again for profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c19
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline > $O/bench_int8_0517_b32.json 2> $O/e1.err
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 300 python bench.py --precision int8 --model mnet25 --global-batch 256 --no-cpu-baseline --host-seconds 0 > $O/bench_int8_mnet25_global256_1gpu.json 2> $O/e4.err
timeout 400 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -3; tail -1 $O/smoke.log
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
