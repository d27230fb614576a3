#!/bin/bash
# round 3, GPU call 9: the evidence set of the round on the final build -- whole suite, residency-check cost, host-frame rates, bench lines for
# every BASELINE config (+ the driver's invocation, + configs[4] as one global batch on one GPU), kernel traces (fp16 1 / 3 lanes, int8 1 lane)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c9
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=5 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
./tools/probes/ptr_attr.bin > $O/ptr_attr.log 2>&1
timeout 120 python tools/probes/host_rate.py 1.5 > $O/host_rate.log 2>&1
timeout 400 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline > $O/bench_int8_0517_b32.json 2> $O/e1.err
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 400 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/e5.err
timeout 300 python bench.py --precision int8 --model mnet25 --global-batch 256 --no-cpu-baseline --host-seconds 0 > $O/bench_int8_mnet25_global256_1gpu.json 2> $O/e4.err
cd /tmp; export TMPDIR=/tmp
for cfg in "fp16 mnet25 8 1" "int8 mnet25 32 1" "fp16 mnet25 8 0"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python $R/bench.py --precision $1 --model $2 --batch $3 --lanes $4 --timed-only --no-cpu-baseline > $O/tr_$1_$4.log 2>&1
  db=$(find $O/tr -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $O/kernel_trace_$1_lanes$4.txt > /dev/null
  rm -rf $O/tr
done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract|int8 front|calibration tool" | tail -6; cat $O/ptr_attr.log; grep images $O/host_rate.log
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
