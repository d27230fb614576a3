#!/bin/bash
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
