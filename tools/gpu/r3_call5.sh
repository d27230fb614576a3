#!/bin/bash
# round 3, GPU call 5: evidence with the fused SSH tail in place -- whole suite, bench lines (fp16 metric point, both int8 configs, 1280x896),
# single-lane kernel traces (fp16, int8)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c5
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 400 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline > $O/bench_int8_0517_b32.json 2> $O/e1.err
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 400 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/e5.err
cd /tmp; export TMPDIR=/tmp
for cfg in "fp16 mnet25 8" "int8 mnet25 32"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$1 -o t -- python $R/bench.py --precision $1 --model $2 --batch $3 --lanes 1 --timed-only --no-cpu-baseline > $O/tr_$1.log 2>&1
  db=$(find $O/tr_$1 -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $O/kernel_trace_lanes1_$1.txt > /dev/null
  rm -rf $O/tr_$1
done
grep -v "compute time" $O/pytest.log | tail -4
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
