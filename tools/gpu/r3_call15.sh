#!/bin/bash
# round 3, GPU call 15: final tree (late store int8 only) -- whole suite, smoke(), the bench lines of the round for profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c15
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 400 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline > $O/bench_int8_0517_b32.json 2> $O/e1.err
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 400 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/e5.err
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -3; tail -1 $O/smoke.log
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
