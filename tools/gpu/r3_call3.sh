#!/bin/bash
# round 3, GPU call 3: new tests (calibration tool end to end, RF_STEM2_DC knob), host-side trace of the synchronous call, evidence bench
# lines for every BASELINE config with the physical roofline, kernel traces (3 lanes / 1 lane) of the default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -s --durations=5 -k "calibration_tool or probe_knob or scattered" > $O/pytest_new.log 2>&1
echo "rc $?" >> $O/pytest_new.log
for b in 8 1; do RF_HOST_TRACE=1 timeout 120 python tools/probes/sync_latency.py $b 1 >> $O/sync_latency.log 2>&1; done
RF_HOST_TRACE=1 timeout 120 python tools/probes/sync_latency.py 8 0 >> $O/sync_latency.log 2>&1
timeout 400 python bench.py > $O/bench_b8_448_fp16.json 2> $O/bench_b8_448_fp16.err; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline > $O/bench_int8_0517_b32.json 2> $O/e1.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_0517_b32.json
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 400 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err; cp gpurun_out/bench_kernels.json $O/kernels_1280x896_b1_fp16.json
timeout 300 python bench.py --precision int8 --model mnet25 --global-batch 256 --no-cpu-baseline --host-seconds 0 > $O/bench_int8_mnet25_global256_1gpu.json 2> $O/e4.err
cd /tmp; export TMPDIR=/tmp
for lanes in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$lanes -o t -- python $R/bench.py --lanes $lanes --timed-only --no-cpu-baseline > $O/tr_$lanes.log 2>&1
  db=$(find $O/tr_$lanes -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $O/bench_b8_448_fp16_kernel_trace_lanes$lanes.txt > /dev/null
  rm -rf $O/tr_$lanes
done
grep -v "compute time" $O/pytest_new.log | tail -5; cat $O/sync_latency.log | grep -v amdgpu.ids
