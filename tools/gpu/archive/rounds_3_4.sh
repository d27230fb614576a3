#!/bin/bash
# The GPU calls of rounds 3 and 4, one shell function per call (round 5 replaced one-file-per-call by tools/gpu/r5.sh <tag> <recipes>; these 57 bodies
# are kept verbatim as the provenance of the profiles/r03_* and r04_* files DESIGN.md cites).  usage: bash tools/gpu/rounds_3_4.sh r4_call33
# Many of them select kernel variants through RF_* probe knobs: since round 5 those exist in the probe build only
# (RETINAFACE_AMD_LIB=retinaface_amd/lib/libretinaface_amd_probe.so).

r3_call1() {
# round 3, GPU call 1: the whole -m gpu suite (new: bit-exact int8 parity, fp16 contract over 208 frames) + single-lane kernel traces
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=10 -s -k "int8_engine_is_bit_exact or fp16_contract" > $O/pytest_new.log 2>&1
echo "new tests rc $?" >> $O/pytest_new.log
timeout 600 python -m pytest tests -m gpu -q --durations=10 --deselect tests/test_gpu_parity.py::test_fp16_contract_over_200_frames_both_models_both_sizes -k "not int8_engine_is_bit_exact" > $O/pytest_rest.log 2>&1
echo "rest rc $?" >> $O/pytest_rest.log
cd /tmp; export TMPDIR=/tmp
for cfg in "int8 mnet25 32" "fp16 mnet25 8"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$1 -o t -- python $R/bench.py --precision $1 --model $2 --batch $3 --lanes 1 --timed-only --no-cpu-baseline > $O/trace_$1.log 2>&1
  db=$(find $O/trace_$1 -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $O/trace_$1_lanes1.txt > /dev/null
  rm -rf $O/trace_$1
done
tail -3 $O/pytest_new.log; tail -3 $O/pytest_rest.log
}

r3_call2() {
# round 3, GPU call 2: suite after the multi-GPU scatter / lazy lanes / plan-cache / stem2 DC-centring changes; A/B of the centring on
# the 208-frame fp16 contract; the reworked bench line (3 regions, physical roofline, sync_batch)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=8 -x --deselect tests/test_gpu_parity.py::test_fp16_contract_over_200_frames_both_models_both_sizes > $O/pytest.log 2>&1
echo "suite rc $?" >> $O/pytest.log
for dc in 1 0; do
  RF_STEM2_DC=$dc timeout 600 python -m pytest tests -m gpu -q -s -k fp16_contract 2>&1 | grep -E "fp16 contract|passed|failed" > $O/contract_dc$dc.log
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $?" >> $O/bench_default.err
cp gpurun_out/bench_kernels.json $O/bench_kernels_fp16.json 2>/dev/null
tail -4 $O/pytest.log; cat $O/contract_dc*.log; head -c 1500 $O/bench_default.json
}

r3_call3() {
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
}

r3_call4() {
# round 3, GPU call 4: the fused SSH tail (ssh_tail_kernel): whole suite, then per-kernel A/B (RF_SSHTAIL = 0 two launches / 1 fused,
# 4 workgroups per CU / 2 fused, 3 per CU) in fp16 and int8 at 256 images per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c4
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for v in 0 1 2; do
  RF_SSHTAIL=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_sshtail$v > $O/kbench_fp16_sshtail$v.txt 2>&1
  RF_SSHTAIL=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_sshtail$v > $O/kbench_int8_sshtail$v.txt 2>&1
done
grep -v "compute time" $O/pytest.log | tail -6; grep -h "==\|ssh_tail\|conv3x3<16" $O/kbench_*.txt
}

r3_call5() {
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
}

r3_call6() {
# round 3, GPU call 6: conv0's 1024 offset folded into the bias (u8x4_to_f16 without the subtraction): suite, then per-kernel A/B against the
# previous build (retinaface_amd/lib_base) inside one call, fp16 and int8, twice each (box drift)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do
  for which in base new; do
    lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
    RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --tag fp16_${which}_$rep > $O/kbench_fp16_${which}_$rep.txt 2>&1
    RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_${which}_$rep > $O/kbench_int8_${which}_$rep.txt 2>&1
  done
done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract|int8 front" | tail -6; grep -h "==\|  stem" $O/kbench_*.txt
}

r3_call7() {
# round 3, GPU call 7: whole suite on the offset-folded conv0 build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c7
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract|int8 front|calibration tool" | tail -8
}

r3_call8() {
# round 3, GPU call 8: A/B of a second upload stream for host frames (RF_COPY_STREAMS), interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c8
mkdir -p $O
cd $R
for rep in 1 2; do for cs in 1 2; do RF_COPY_STREAMS=$cs timeout 120 python tools/probes/host_rate.py 1.5 >> $O/host_rate.log 2>&1; done; done
grep images $O/host_rate.log
}

r3_call9() {
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
}

r3_call10() {
# round 3, GPU call 10: co-scheduling experiment -- stem2 capped at 7 / 6 workgroups per CU (RF_STEM2_PAD = 3 / 7 KB of unused LDS) so that another
# lane's memory-bound kernels can run beside it; three-lane pipeline throughput, interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c10
mkdir -p $O
cd $R
for rep in 1 2; do for pad in 0 3 7; do
  RF_STEM2_PAD=$pad timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pad $pad rep $rep lanes3', round(j['images_per_sec']), [round(x) for x in j['regions']['faces_per_sec']])" >> $O/pad.log
done; done
for pad in 0 7; do RF_STEM2_PAD=$pad timeout 200 python bench.py --timed-only --no-cpu-baseline --lanes 4 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pad $pad lanes4', round(j['images_per_sec']))" >> $O/pad.log; done
cat $O/pad.log
}

r3_call11() {
# round 3, GPU call 11: dwpw2 at 4 workgroups per CU (128 VGPRs: 3 + 2 unit split, biases read from LDS): fp16 tests, then A/B against the previous
# build (per-kernel and three-lane pipeline), interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c11
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "not int8 and not calibration" -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do for which in base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --tag fp16_${which}_$rep > $O/kbench_fp16_${which}_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done; done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; grep -h "==\|dwpw2" $O/kbench_*.txt; cat $O/pipe.log
}

r3_call12() {
# round 3, GPU call 12: lanes sweep for the int8 engine (its kernels leave more of the chip idle than the fp16 ones: three lanes give +14 % over
# the single-lane kernel sum, against +5 % in fp16), interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c12
mkdir -p $O
cd $R
for rep in 1 2; do for lanes in 3 4 5 6; do
  timeout 200 python bench.py --precision int8 --model mnet25 --batch 32 --lanes $lanes --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('int8 lanes $lanes rep $rep', round(j['images_per_sec']))" >> $O/lanes.log
done; done
cat $O/lanes.log
}

r3_call13() {
# round 3, GPU call 13: fewer barriers per tile (dwpw without lateral: 3 -> 2; conv3x3 double buffered: 2 -> 1; ssh_tail: 3 -> 2): whole suite
# (twice for the determinism / batch-invariance tests: a missing barrier is a race), then A/B against the previous build, fp16 + int8, per kernel
# and three-lane pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c13
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 600 python -m pytest tests -m gpu -q -x -k "determinism or batch_composition or bit_exact or synthetic_batch8 or every_fused_op" > $O/pytest_again.log 2>&1
echo "rc $?" >> $O/pytest_again.log
for rep in 1 2; do for which in base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --tag fp16_${which}_$rep > $O/kbench_fp16_${which}_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_${which}_$rep > $O/kbench_int8_${which}_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep fp16 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --precision int8 --batch 32 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep int8 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done; done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; tail -2 $O/pytest_again.log; grep -h "==" $O/kbench_*.txt; cat $O/pipe.log
}

r3_call14() {
# round 3, GPU call 14: final tree -- whole suite, smoke(), three-lane pipeline A/B against the build before the barrier changes (fp16, int8),
# the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c14
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
for rep in 1 2; do for which in base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep fp16 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --precision int8 --batch 32 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep int8 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done; done
timeout 400 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -3; tail -2 $O/smoke.log; cat $O/pipe.log
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
}

r3_call15() {
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
}

r3_call16() {
# round 3, GPU call 16 (probe): does v_cvt_pk_u8_f32 round to nearest even by itself?  The bit-exact int8 test on a build without v_rndne_f32
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c16
mkdir -p $O
cd $R
RETINAFACE_AMD_LIB=$R/retinaface_amd/lib_dev/libretinaface_amd.so timeout 600 python -m pytest tests -m gpu -q -x -k "int8_engine_is_bit_exact" > $O/pytest_no_rndne.log 2>&1
grep -v "compute time" $O/pytest_no_rndne.log | grep -E "passed|failed|AssertionError|assert " | head -6
}

r3_call18() {
# round 3, GPU call 18: int8 epilogues without the separate v_rndne_f32 (v_cvt_pk_u8_f32 rounds to nearest even itself): int8 tests, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c18
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "int8 or calibration" > $O/pytest_int8.log 2>&1
echo "rc $?" >> $O/pytest_int8.log
for rep in 1 2; do for which in base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_${which}_$rep > $O/kbench_int8_${which}_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --precision int8 --batch 32 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep int8 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done; done
grep -v "compute time" $O/pytest_int8.log | tail -3; grep -h "==\|  stem" $O/kbench_*.txt; cat $O/pipe.log
}

r3_call19() {
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
}

r3_call20() {
# round 3, GPU call 20: int8 stem with 1 / scale folded into its pointwise weights (no fma in the epilogue): int8 tests, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c20
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -s -k "int8 or calibration" > $O/pytest_int8.log 2>&1
echo "rc $?" >> $O/pytest_int8.log
for which in base new base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_$which 2>&1 | grep "==\|  stem" >> $O/kbench.log
done
grep -v "compute time" $O/pytest_int8.log | grep -E "passed|failed|front end|Error" | tail -5; cat $O/kbench.log
}

r3_call21() {
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
}

r4_call0() {
# round 4, GPU call 0: LDS-DMA semantics probe (what the chained depthwise/pointwise kernel's staging will rely on) + per-kernel baselines of HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c0
mkdir -p $O
cd $R
timeout 60 tools/probes/lds_dma.bin > $O/lds_dma.txt 2>&1; echo "rc $?" >> $O/lds_dma.txt
timeout 200 python tools/kbench.py --n 256 --tag r4c0_fp16 > $O/kbench_fp16.txt 2>&1
timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r4c0_int8 > $O/kbench_int8.txt 2>&1
cat $O/lds_dma.txt; grep -h "==" $O/kbench_*.txt
}

r4_call1() {
# round 4, GPU call 1: phase timelines (s_memtime stamps, probe build) of the kernels the verdict names as latency bound, at 256 images per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c1
mkdir -p $O
cd $R
timeout 120 python tools/probes/phase_trace.py conv3 256 1072 25152 6336 > $O/trace_conv3.txt 2>&1
timeout 120 python tools/probes/phase_trace.py dwpw 256 7168 25088 > $O/trace_dwpw.txt 2>&1
timeout 120 python tools/probes/phase_trace.py dwpw2 256 0 > $O/trace_dwpw2.txt 2>&1
timeout 120 python tools/probes/phase_trace.py stem2 256 0 > $O/trace_stem2.txt 2>&1
tail -n 30 $O/trace_conv3.txt $O/trace_dwpw.txt $O/trace_dwpw2.txt $O/trace_stem2.txt
}

r4_call2() {
# round 4, GPU call 2: warp-specialised SSH conv (conv3x3_ws_kernel): parity subset (per-op blobs, goldens, bit-exact int8, determinism, 1280x896, odd sizes),
# then A/B against RF_CONV3WS=0 inside the same call, fp16 + int8, per kernel and three-lane pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or fixture_image or bit_exact or determinism or batch_composition or odd_net_size or edge_cases or knob" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do for ws in 0 1; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_ws${ws}_$rep > $O/kbench_fp16_ws${ws}_$rep.txt 2>&1
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_ws${ws}_$rep > $O/kbench_int8_ws${ws}_$rep.txt 2>&1
done; done
for ws in 0 1; do
  RF_CONV3WS=$ws timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ws $ws fp16 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
  RF_CONV3WS=$ws timeout 200 python bench.py --precision int8 --batch 32 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ws $ws int8 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done
grep -v "compute time" $O/pytest.log | tail -5; grep -h "==\|conv3x3<64,48" $O/kbench_*.txt; cat $O/pipe.log
}

r4_call3() {
# round 4, GPU call 3: B-fragment software pipeline pinned with sched_barrier in gemm_stationary (the compiler had sunk every ds_read to its MFMA):
# parity subset, then A/B against the round-3 library (lib_base), fp16 + int8, WS conv on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do
  RETINAFACE_AMD_LIB=$R/retinaface_amd/lib_base/libretinaface_amd.so timeout 200 python tools/kbench.py --n 256 --tag fp16_base_$rep > $O/kbench_fp16_base_$rep.txt 2>&1
  RF_CONV3WS=0 timeout 200 python tools/kbench.py --n 256 --tag fp16_pin_ws0_$rep > $O/kbench_fp16_pin_ws0_$rep.txt 2>&1
  RF_CONV3WS=1 timeout 200 python tools/kbench.py --n 256 --tag fp16_pin_ws1_$rep > $O/kbench_fp16_pin_ws1_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$R/retinaface_amd/lib_base/libretinaface_amd.so timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_base_$rep > $O/kbench_int8_base_$rep.txt 2>&1
  RF_CONV3WS=0 timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_pin_ws0_$rep > $O/kbench_int8_pin_ws0_$rep.txt 2>&1
  RF_CONV3WS=1 timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_pin_ws1_$rep > $O/kbench_int8_pin_ws1_$rep.txt 2>&1
done
grep -v "compute time" $O/pytest.log | tail -3; grep -h "==" $O/kbench_*.txt
}

r4_call4() {
# round 4, GPU call 4: warp-specialised SSH conv with the pinned pipeline at prefetch depth 2 / 3 / 4 vs the lock-step kernel (pinned too), 3 repetitions, fp16 + int8
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "every_fused_op or bit_exact or determinism" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for ws in 0 2 3 4; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_ws${ws}_$rep > $O/kbench_fp16_ws${ws}_$rep.txt 2>&1
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_ws${ws}_$rep > $O/kbench_int8_ws${ws}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') $(grep -h 'conv3x3<64,48' $f | awk '{print $2}') $(grep -h 'ssh_tail' $f | awk '{print $2}')"; done
}

r4_call5() {
# round 4, GPU call 5: warp-specialised SSH conv with 2 or 3 halo buffers (producer 1 or 2 tiles ahead, counted vmcnt) x prefetch depth 2 / 3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c5
mkdir -p $O
cd $R
for v in 22 32 33; do
  RF_CONV3WS=$v timeout 600 python -m pytest tests -m gpu -q -x -k "every_fused_op or bit_exact or determinism or odd_net_size" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3 4; do for ws in 0 22 23 32 33; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_ws${ws}_$rep > $O/kbench_fp16_ws${ws}_$rep.txt 2>&1
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_ws${ws}_$rep > $O/kbench_int8_ws${ws}_$rep.txt 2>&1
done; done
for v in 22 32 33; do grep -v "compute time" $O/pytest_$v.log | tail -2; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') $(grep -h 'conv3x3<64,48' $f | awk '{print $2}')"; done
}

r4_call6() {
# round 4, GPU call 6: the tightened fp16 bands / order / per-layer bars on the GPU, and the new default bench line (configs array, physical frac)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -s -k "every_fused_op or golden or fixture_image or fp16_contract" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
grep -v "compute time" $O/pytest.log | grep -E "fp16 layer|fp16 contract|passed|failed|Error|assert" | tail -40
tail -3 $O/bench_time.txt; tail -5 $O/bench_default.err
python - <<'P'
import json
j=json.loads(open('/root/repo/gpurun_out/r4c6/bench_default.json').read().strip().splitlines()[-1])
print(round(j['images_per_sec']), j['roofline']['bound'], j['roofline']['frac'], j['roofline']['unit'])
for c in j.get('configs', []): print(c['id'], round(c['images_per_sec']), round(c['faces_per_sec']), c['dominant_kernel'], c.get('bound'), c.get('bound_frac'), c.get('hbm_frac_measured'))
print(j.get('cpu_baseline',{}).get('value'))
P
}

r4_call7() {
# round 4, GPU call 7: warp-specialised depthwise/pointwise blocks (dwpw_ws_kernel, 64- and 128-channel stride-1 blocks +- lateral): parity with 2 and 3 halo
# buffers (incl. the bit-exact int8 suite), then A/B per kernel against K_b
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c7
mkdir -p $O
cd $R
for v in 2 3; do
  RF_DWPWWS=$v timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3; do for ws in 0 2 3; do
  RF_DWPWWS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_dws${ws}_$rep > $O/kbench_fp16_dws${ws}_$rep.txt 2>&1
  RF_DWPWWS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_dws${ws}_$rep > $O/kbench_int8_dws${ws}_$rep.txt 2>&1
done; done
for v in 2 3; do grep -v "compute time" $O/pytest_$v.log | tail -3; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw<64,64,s1,lat>\|dwpw<128,128' $f | awk '{printf "%s ", $2}')"; done
}

r4_call8() {
# round 4, GPU call 8: the whole -m gpu suite on the tree with the ADVICE fixes, the rehearsal tests and the int8 self-check; smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c8
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
grep -v "compute time" $O/pytest.log | tail -15; tail -2 $O/smoke.log
}

r4_call9() {
# round 4, GPU call 9: warp-specialised aggregation convs (conv3x3_up_ws_kernel: LDS-DMA of lateral halo + coarse patch, blend LDS -> LDS, 8x8 tiles):
# parity with 2 and 3 ring buffers (incl. bit-exact int8 and the integer-blend test), then A/B per kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c9
mkdir -p $O
cd $R
for v in 2 3; do
  RF_CONV3UPWS=$v timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image or integer_blend or edge_cases" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3; do for ws in 0 2 3; do
  RF_CONV3UPWS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_up${ws}_$rep > $O/kbench_fp16_up${ws}_$rep.txt 2>&1
  RF_CONV3UPWS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_up${ws}_$rep > $O/kbench_int8_up${ws}_$rep.txt 2>&1
done; done
for v in 2 3; do grep -v "compute time" $O/pytest_$v.log | tail -3; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'conv3x3<64,64' $f | awk '{printf "%s ", $2}')"; done
}

r4_call10() {
# round 4, GPU call 10: producer joins the mid-interval barrier before its issue work (UPADD WS conv), DMA after the first barrier (dwpw WS): A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c10
mkdir -p $O
cd $R
RF_CONV3UPWS=3 RF_DWPWWS=3 timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or integer_blend" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do for ws in 0 2 3; do
  RF_CONV3UPWS=$ws RF_DWPWWS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_v${ws}_$rep > $O/kbench_fp16_v${ws}_$rep.txt 2>&1
  RF_CONV3UPWS=$ws RF_DWPWWS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_v${ws}_$rep > $O/kbench_int8_v${ws}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'conv3x3<64,64\|dwpw<64,64,s1,lat>\|dwpw<128,128' $f | awk '{printf "%s ", $2}')"; done
}

r4_call11() {
# round 4, GPU call 11: why is the warp-specialised aggregation conv slow?  phase stamps of consumer wave 0 and the producer wave; dwpw2 ring A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c11
mkdir -p $O
cd $R
RF_CONV3UPWS=3 timeout 200 python tools/probes/ws_trace.py 256 > $O/ws_trace_nbuf3.txt 2>&1
RF_CONV3UPWS=2 timeout 200 python tools/probes/ws_trace.py 256 > $O/ws_trace_nbuf2.txt 2>&1
for rep in 1 2 3; do for r in 0 1; do
  RF_DWPW2_RING=$r timeout 200 python tools/kbench.py --n 256 --tag fp16_ring${r}_$rep > $O/kbench_fp16_ring${r}_$rep.txt 2>&1
done; done
RF_DWPW2_RING=1 timeout 600 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or determinism" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
cat $O/ws_trace_nbuf3.txt $O/ws_trace_nbuf2.txt | grep -v amdgpu.ids
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw2' $f | awk '{printf "%s ", $2}')"; done; tail -3 $O/pytest.log
}

r4_call12() {
# round 4, GPU call 12: aggregation convs with LDS-DMA staging and every wave issuing its own share (no producer wave): parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c12
mkdir -p $O
cd $R
for v in 12 13; do
  RF_CONV3UPWS=$v timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image or integer_blend or edge_cases" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3; do for ws in 0 12 13; do
  RF_CONV3UPWS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_up${ws}_$rep > $O/kbench_fp16_up${ws}_$rep.txt 2>&1
  RF_CONV3UPWS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_up${ws}_$rep > $O/kbench_int8_up${ws}_$rep.txt 2>&1
done; done
for v in 12 13; do grep -v "compute time" $O/pytest_$v.log | tail -3; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'conv3x3<64,64' $f | awk '{printf "%s ", $2}')"; done
}

r4_call13() {
# round 4, GPU call 13: 8x8 tiles for the 256-channel block (the streamed weight matrix is read once per 64 pixels instead of 32): parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c13
mkdir -p $O
cd $R
RF_TILE256=1 timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  RF_TILE256=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_t256_${v}_$rep > $O/kbench_fp16_t256_${v}_$rep.txt 2>&1
  RF_TILE256=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_t256_${v}_$rep > $O/kbench_int8_t256_${v}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw<256,256' $f | awk '{printf "%s ", $2}')"; done
}

r4_call14() {
# round 4, GPU call 14: warp-specialised SSH conv with the 2 + 2 + 1 role split of the GEMM waves (LDS B-fragment traffic 216 -> 144 KB per tile)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c14
mkdir -p $O
cd $R
for v in 132; do
  RF_CONV3WS=$v timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3; do for ws in 1 132 122; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_ws${ws}_$rep > $O/kbench_fp16_ws${ws}_$rep.txt 2>&1
done; for ws in 0 132 122; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_ws${ws}_$rep > $O/kbench_int8_ws${ws}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest_132.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'conv3x3<64,48' $f | awk '{printf "%s ", $2}')"; done
}

r4_call15() {
# round 4, GPU call 15: the tree with the round's defaults -- whole suite, smoke(), the default bench line (all BASELINE configs), per-config lines,
# single-lane rocprofv3 kernel traces (fp16 b8, int8 b32) and per-kernel tables
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c15
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time timeout 600 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err ) 2> $O/bench_time.txt; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/e5.err
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline --no-pmc > $O/bench_int8_0517_b32.json 2> $O/e1.err
timeout 400 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline --no-pmc > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err
timeout 200 python tools/kbench.py --n 256 --tag r4c15_fp16 > $O/kbench_fp16.txt 2>&1
timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r4c15_int8 > $O/kbench_int8.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_fp16 $O/trace_int8
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 > $O/trace_fp16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_int8 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 --precision int8 --batch 32 > $O/trace_int8.log 2>&1
cd $R
for t in fp16 int8; do db=$(find $O/trace_$t -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_$t.txt > /dev/null; rm -rf $O/trace_$t; done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; tail -1 $O/smoke.log; tail -3 $O/bench_time.txt
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
grep -h "==" $O/kbench_*.txt; head -12 $O/kernel_trace_lanes1_fp16.txt | cut -c1-60,100-190
}

r4_call16() {
# round 4, GPU call 16: tile shapes of the int8 64- and 128-channel blocks (8x8, 4x16, 8x16 instead of 4x8): bit-identity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c16
mkdir -p $O
cd $R
timeout 900 python tools/probes/knob_equal.py --precision 2 RF_TILE128=1 RF_TILE128=2 RF_TILE128=3 RF_TILE64=1 RF_TILE64=2 > $O/equal_int8.txt 2>&1
timeout 600 python tools/probes/knob_equal.py --precision 1 RF_TILE128=1 RF_TILE128=2 RF_TILE64=1 > $O/equal_fp16.txt 2>&1
for rep in 1 2; do
  for v in 0 1 2 3; do RF_TILE128=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_t128_${v}_$rep > $O/kbench_int8_t128_${v}_$rep.txt 2>&1; done
  for v in 1 2; do RF_TILE64=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_t64_${v}_$rep > $O/kbench_int8_t64_${v}_$rep.txt 2>&1; done
  for v in 0 1 2; do RF_TILE128=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_t128_${v}_$rep > $O/kbench_fp16_t128_${v}_$rep.txt 2>&1; done
  RF_TILE64=1 timeout 200 python tools/kbench.py --n 256 --tag fp16_t64_1_$rep > $O/kbench_fp16_t64_1_$rep.txt 2>&1
done
cat $O/equal_int8.txt $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw<64,64\|dwpw<128,128' $f | awk '{printf "%s ", $2}')"; done
}

r4_call17() {
# round 4, GPU call 17: the library compiled with -mllvm --amdgpu-mfma-vgpr-form (MFMA results land in VGPRs: no v_accvgpr_read copies in the epilogues): identity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c17
mkdir -p $O
cd $R
VF=$R/retinaface_amd/lib_vf/libretinaface_amd.so
timeout 600 python tools/probes/knob_equal.py --precision 2 RETINAFACE_AMD_LIB=$VF > $O/equal_int8.txt 2>&1
timeout 600 python tools/probes/knob_equal.py --precision 1 RETINAFACE_AMD_LIB=$VF > $O/equal_fp16.txt 2>&1
for rep in 1 2 3; do
  timeout 200 python tools/kbench.py --n 256 --tag fp16_base_$rep > $O/kbench_fp16_base_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$VF timeout 200 python tools/kbench.py --n 256 --tag fp16_vf_$rep > $O/kbench_fp16_vf_$rep.txt 2>&1
  timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_base_$rep > $O/kbench_int8_base_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$VF timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_vf_$rep > $O/kbench_int8_vf_$rep.txt 2>&1
done
cat $O/equal_int8.txt $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -v '==' $f | grep us | awk '{printf "%s ", $2}')"; done
}

r4_call18() {
# round 4, GPU call 18: larger tiles for the int8 engine's big-map blocks (16->32 s2, 32->32, 32->64 s2, 64->128 s2): bit-identity + A/B; library now built in VGPR-form
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c18
mkdir -p $O
cd $R
timeout 900 python tools/probes/knob_equal.py --precision 2 RF_TILE_A=1 RF_TILE_A=2 RF_TILE_B=1 RF_TILE_B=2 RF_TILE_C=1 RF_TILE_C=2 RF_TILE_D=1 RF_TILE128=2 > $O/equal_int8.txt 2>&1
for rep in 1 2; do
  timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_base_$rep > $O/kbench_int8_base_$rep.txt 2>&1
  for k in RF_TILE_A=1 RF_TILE_A=2 RF_TILE_B=1 RF_TILE_B=2 RF_TILE_C=1 RF_TILE_C=2 RF_TILE_D=1 RF_TILE128=2; do
    env $k timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_${k/=/_}_$rep > $O/kbench_int8_${k/=/_}_$rep.txt 2>&1
  done
done
cat $O/equal_int8.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -v '==' $f | grep us | awk '{printf "%s ", $2}')"; done
}

r4_call19() {
# round 4, GPU call 19: the tree with the round's defaults -- whole suite, smoke(), the default bench line (all BASELINE configs), per-config lines,
# single-lane rocprofv3 kernel traces (fp16 b8, int8 b32) and per-kernel tables
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c19
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time timeout 600 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err ) 2> $O/bench_time.txt; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/e5.err
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline --no-pmc > $O/bench_int8_0517_b32.json 2> $O/e1.err
timeout 400 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline --no-pmc > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err
timeout 200 python tools/kbench.py --n 256 --tag r4c19_fp16 > $O/kbench_fp16.txt 2>&1
timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r4c19_int8 > $O/kbench_int8.txt 2>&1
for l in 2 3 4; do timeout 200 python bench.py --timed-only --no-cpu-baseline --lanes $l --min-seconds 0.5 > $O/lanes_${l}_fp16.json 2>/dev/null; timeout 200 python bench.py --timed-only --no-cpu-baseline --lanes $l --min-seconds 0.5 --precision int8 --batch 32 > $O/lanes_${l}_int8.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_fp16 $O/trace_int8
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 > $O/trace_fp16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_int8 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 --precision int8 --batch 32 > $O/trace_int8.log 2>&1
cd $R
for t in fp16 int8; do db=$(find $O/trace_$t -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_$t.txt > /dev/null; rm -rf $O/trace_$t; done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; tail -1 $O/smoke.log; tail -3 $O/bench_time.txt
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
grep -h "==" $O/kbench_*.txt; head -12 $O/kernel_trace_lanes1_fp16.txt | cut -c1-60,100-190
for f in $O/lanes_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['images_per_sec']))"; done
}

r4_call20() {
# round 4, GPU call 20: one synchronous call of 8 / 1 device-resident frames: hipGraphLaunch vs eager launches (host trace of both)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c20
mkdir -p $O
cd $R
for b in 8 1; do for g in 1 0; do
  RF_HOST_TRACE=1 timeout 200 python tools/probes/sync_latency.py $b $g > $O/sync_b${b}_g${g}.txt 2>&1
done; done
tail -n 12 $O/sync_*.txt
}

r4_call21() {
# round 4, GPU call 21: K_b'' -- the 64- / 128-channel depthwise-pointwise blocks with the halo DMA and the stores spread over the four GEMM waves
# (RF_DWPWWS=12 / 13: 2 / 3 halo buffers): identity + A/B against K_b
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c21
mkdir -p $O
cd $R
timeout 600 python tools/probes/knob_equal.py --precision 2 RF_DWPWWS=12 RF_DWPWWS=13 > $O/equal_int8.txt 2>&1
timeout 600 python tools/probes/knob_equal.py --precision 1 RF_DWPWWS=12 RF_DWPWWS=13 > $O/equal_fp16.txt 2>&1
for rep in 1 2; do for v in 0 12 13; do
  RF_DWPWWS=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_dd${v}_$rep > $O/kbench_fp16_dd${v}_$rep.txt 2>&1
  RF_DWPWWS=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_dd${v}_$rep > $O/kbench_int8_dd${v}_$rep.txt 2>&1
done; done
cat $O/equal_int8.txt $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw<64,64\|dwpw<128,128' $f | awk '{printf "%s ", $2}')"; done
}

r4_call22() {
# round 4, GPU call 22: LDS data-path counters of every kernel (is LDS bandwidth / bank conflicts what bounds the depthwise-pointwise blocks?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c22
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_LDS\|SQ_WAIT_INST_LDS\|SQ_INST_CYCLES_[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\b\|SQ_ACTIVE_INST_[A-Z_]*" | sort -u > $O/counters_available.txt
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $O/p_$tag
  timeout 300 rocprofv3 --pmc $set GRBM_GUI_ACTIVE -d $O/p_$tag -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p_$tag.log 2>&1
  db=$(find $O/p_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/probes/lds_counters.py $db $set > $O/lds_$tag.txt 2>&1
  rm -rf $O/p_$tag
done
cat $O/counters_available.txt | tr '\n' ' '; echo; cat $O/lds_*.txt | cut -c1-200
}

r4_call23() {
# round 4, GPU call 23: dwpw2 with padded halo rows (bank-conflict-free depthwise-A reads): parity subset, A/B, LDS counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c23
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or determinism or odd_net_size or fixture_image" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  RF_DWPW2_HPAD=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_hpad${v}_$rep > $O/kbench_fp16_hpad${v}_$rep.txt 2>&1
done; done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $O/p_$v
  RF_DWPW2_HPAD=$v timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p_$v -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p_$v.log 2>&1
  db=$(find $O/p_$v -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_hpad$v.txt 2>&1
  rm -rf $O/p_$v
done
cd $R
grep -v "compute time" $O/pytest.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw2' $f | awk '{printf "%s ", $2}')"; done
grep -h "kernel \|dwpw2" $O/lds_hpad*.txt | cut -c1-150
}

r4_call24() {
# round 4, GPU call 24: dwpw2 with the depthwise -> pointwise hops chained in registers (RF_DWPW2_CHAIN=1) vs K_b2 with padded halo rows: parity subset + A/B + LDS counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c24
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or determinism or odd_net_size or fixture_image" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  RF_DWPW2_CHAIN=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_chain${v}_$rep > $O/kbench_fp16_chain${v}_$rep.txt 2>&1
done; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p_1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p_1 -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p_1.log 2>&1
db=$(find $O/p_1 -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_chain1.txt 2>&1
rm -rf $O/p_1
cd $R
grep -v "compute time" $O/pytest.log | tail -12
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw2' $f | awk '{printf "%s ", $2}')"; done
grep -h "kernel \|dwpw2" $O/lds_chain1.txt | cut -c1-150
}

r4_call25() {
# round 4, GPU call 25: dwpw2 with the depthwise -> pointwise hops chained in registers (RF_DWPW2_CHAIN=1) vs K_b2 with padded halo rows: parity subset + A/B + LDS counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c25
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or determinism or odd_net_size or fixture_image" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  RF_DWPW2_CHAIN=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_chain${v}_$rep > $O/kbench_fp16_chain${v}_$rep.txt 2>&1
done; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p_1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p_1 -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p_1.log 2>&1
db=$(find $O/p_1 -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_chain1.txt 2>&1
rm -rf $O/p_1
cd $R
grep -v "compute time" $O/pytest.log | tail -12
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw2' $f | awk '{printf "%s ", $2}')"; done
grep -h "kernel \|dwpw2" $O/lds_chain1.txt | cut -c1-150
}

r4_call26() {
# round 4, GPU call 26: K_b2 with the conflict-reducing LDS layouts (RF_DWPW2_LAY2=1: 80-byte pitches for the depthwise-A and block-A tiles, 2-D block-A tile) vs the 96-byte ones
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c26
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or determinism or odd_net_size or fixture_image" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  RF_DWPW2_LAY2=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_lay${v}_$rep > $O/kbench_fp16_lay${v}_$rep.txt 2>&1
done; done
timeout 300 python tools/probes/knob_equal.py --precision 1 RF_DWPW2_LAY2=0 RF_DWPW2_HPAD=0 > $O/equal_fp16.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p_1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p_1 -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p_1.log 2>&1
db=$(find $O/p_1 -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_lay1.txt 2>&1
rm -rf $O/p_1
cd $R
grep -v "compute time" $O/pytest.log | tail -4; cat $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw2' $f | awk '{printf "%s ", $2}')"; done
grep -h "kernel \|dwpw2" $O/lds_lay1.txt | cut -c1-150
}

r4_call27() {
# round 4, GPU call 27: LDS data-path counters of the int8 engine's kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c27
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p -o pmc -- python $R/tools/probes/pmc_probe.py 256 int8 mnet25 448 448 32 > $O/p.log 2>&1
db=$(find $O/p -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_int8.txt 2>&1
rm -rf $O/p
cut -c1-150 $O/lds_int8.txt
}

r4_call28() {
# round 4, GPU call 28: the tree with the round's defaults -- whole suite, smoke(), the default bench line (all BASELINE configs), per-config lines,
# single-lane rocprofv3 kernel traces (fp16 b8, int8 b32) and per-kernel tables
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c28
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time timeout 600 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err ) 2> $O/bench_time.txt; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/e5.err
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline --no-pmc > $O/bench_int8_0517_b32.json 2> $O/e1.err
timeout 400 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline --no-pmc > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err
timeout 200 python tools/kbench.py --n 256 --tag r4c28_fp16 > $O/kbench_fp16.txt 2>&1
timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r4c28_int8 > $O/kbench_int8.txt 2>&1
for l in 2 3 4; do timeout 200 python bench.py --timed-only --no-cpu-baseline --lanes $l --min-seconds 0.5 > $O/lanes_${l}_fp16.json 2>/dev/null; timeout 200 python bench.py --timed-only --no-cpu-baseline --lanes $l --min-seconds 0.5 --precision int8 --batch 32 > $O/lanes_${l}_int8.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_fp16 $O/trace_int8
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 > $O/trace_fp16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_int8 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 --precision int8 --batch 32 > $O/trace_int8.log 2>&1
rm -rf $O/p_lds
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p_lds -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p_lds.log 2>&1
db=$(find $O/p_lds -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_counters_fp16.txt 2>&1
rm -rf $O/p_lds
cd $R
for t in fp16 int8; do db=$(find $O/trace_$t -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_$t.txt > /dev/null; rm -rf $O/trace_$t; done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; tail -1 $O/smoke.log; tail -3 $O/bench_time.txt
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
grep -h "==" $O/kbench_*.txt; head -12 $O/kernel_trace_lanes1_fp16.txt | cut -c1-60,100-190
for f in $O/lanes_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['images_per_sec']))"; done
}

r4_call29() {
# round 4, GPU call 29: stem2 V2 (planar conv2 tile: 4-way -> 2-way epilogue writes; conv3 -> conv4 chained in registers: one tile and one barrier less): parity subset + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c29
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or determinism or odd_net_size or fixture_image or contract" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_s2v${v}_$rep > $O/kbench_fp16_s2v${v}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest.log | tail -5
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'stem2' $f | awk '{printf "%s ", $2}')"; done
}

r4_call30() {
# round 4, GPU call 30: stem2 V2 bits: 1 = planar conv2 tile (layout only), 2 = conv3 -> conv4 chained in registers, 3 = both: identity of (1), A/B of all
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c30
mkdir -p $O
cd $R
timeout 300 python tools/probes/knob_equal.py --precision 1 RF_STEM2_V2=0 RF_STEM2_V2=2 RF_STEM2_V2=3 > $O/equal_fp16.txt 2>&1
for rep in 1 2 3; do for v in 0 1 2 3; do
  RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_s2v${v}_$rep > $O/kbench_fp16_s2v${v}_$rep.txt 2>&1
done; done
cat $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'stem2' $f | awk '{printf "%s ", $2}')"; done
}

r4_call31() {
# round 4, GPU call 31: the tree with the round's defaults -- whole suite, smoke(), the default bench line (all BASELINE configs), per-config lines,
# single-lane rocprofv3 kernel traces (fp16 b8, int8 b32) and per-kernel tables
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c31
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time timeout 600 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err ) 2> $O/bench_time.txt; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/e5.err
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline --no-pmc > $O/bench_int8_0517_b32.json 2> $O/e1.err
timeout 400 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline --no-pmc > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err
timeout 200 python tools/kbench.py --n 256 --tag r4c31_fp16 > $O/kbench_fp16.txt 2>&1
timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r4c31_int8 > $O/kbench_int8.txt 2>&1
for l in 2 3 4; do timeout 200 python bench.py --timed-only --no-cpu-baseline --lanes $l --min-seconds 0.5 > $O/lanes_${l}_fp16.json 2>/dev/null; timeout 200 python bench.py --timed-only --no-cpu-baseline --lanes $l --min-seconds 0.5 --precision int8 --batch 32 > $O/lanes_${l}_int8.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_fp16 $O/trace_int8
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 > $O/trace_fp16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_int8 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 --precision int8 --batch 32 > $O/trace_int8.log 2>&1
rm -rf $O/p_lds
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p_lds -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p_lds.log 2>&1
db=$(find $O/p_lds -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_counters_fp16.txt 2>&1
rm -rf $O/p_lds
cd $R
for t in fp16 int8; do db=$(find $O/trace_$t -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_$t.txt > /dev/null; rm -rf $O/trace_$t; done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; tail -1 $O/smoke.log; tail -3 $O/bench_time.txt
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
grep -h "==" $O/kbench_*.txt; head -12 $O/kernel_trace_lanes1_fp16.txt | cut -c1-60,100-190
for f in $O/lanes_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(j['images_per_sec']))"; done
}

r4_call32() {
# round 4, GPU call 32: stem2 V2 bit 2 = rotated thread -> pixel map of the depthwise-1 phase (conflict-free tap reads): identity + A/B + LDS counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c32
mkdir -p $O
cd $R
timeout 300 python tools/probes/knob_equal.py --precision 1 RF_STEM2_V2=5 > $O/equal_fp16.txt 2>&1
for rep in 1 2 3; do for v in 1 5; do
  RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_s2v${v}_$rep > $O/kbench_fp16_s2v${v}_$rep.txt 2>&1
done; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p
RF_STEM2_V2=5 timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p.log 2>&1
db=$(find $O/p -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_s2v5.txt 2>&1
rm -rf $O/p
cd $R
cat $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'stem2' $f | awk '{printf "%s ", $2}')"; done
grep -h "kernel \|stem2" $O/lds_s2v5.txt | cut -c1-150
}

r4_call33() {
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
}

r4_call34() {
# round 4, GPU call 34: the final tree's single-lane fp16 kernel trace and the 1280x896 line (the rest of the final evidence: r4_call33.sh; int8, untouched since: r4_call31.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c34
mkdir -p $O
cd $R
timeout 200 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline --no-pmc --no-extra-configs > $O/bench_1280x896_b1_fp16.json 2> $O/e3.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_fp16
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o t -- python $R/bench.py --timed-only --no-cpu-baseline --lanes 1 --min-seconds 0.5 --regions 1 > $O/trace_fp16.log 2>&1
cd $R
db=$(find $O/trace_fp16 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_fp16.txt > /dev/null; rm -rf $O/trace_fp16
python -c "
import json; j=json.loads(open('$O/bench_1280x896_b1_fp16.json').read().strip().splitlines()[-1]); print(round(j['images_per_sec']), round(j['value']))"
head -8 $O/kernel_trace_lanes1_fp16.txt | cut -c1-60,100-190
}

r4_call35() {
# round 4, GPU call 35: stem2 RF_STEM2_V2 = 5 (default) vs 7 (+ conv3 -> conv4 chained in registers): what the chain would add on top of the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c35
mkdir -p $O
cd $R
for rep in 1 2 3; do for v in 5 7; do
  RF_STEM2_V2=$v timeout 100 python tools/kbench.py --n 256 --tag fp16_s2v${v}_$rep > $O/kbench_fp16_s2v${v}_$rep.txt 2>&1
done; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'stem2' $f | awk '{printf "%s ", $2}')"; done
}

r4_call36() {
# round 4, GPU call 36: LDS data-path counters of every fp16 kernel of the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c36
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p
timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/p.log 2>&1
db=$(find $O/p -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS > $O/lds_counters_fp16.txt 2>&1
rm -rf $O/p
cut -c1-150 $O/lds_counters_fp16.txt | head -16
}

if [ -z "$1" ] || ! declare -F "$1" > /dev/null; then echo "usage: $0 <call>; calls:"; declare -F | awk '{print $3}' | tr "\n" " "; echo; exit 1; fi
"$@"
