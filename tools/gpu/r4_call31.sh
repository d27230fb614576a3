#!/bin/bash
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
