#!/bin/bash
# Everything the round's DESIGN.md / profiles/ numbers come from, in one gpurun call (from the repo root on the GPU box):
#   tools/round_evidence.sh r02
# bench lines for every BASELINE.json config, the rocprofv3 kernel trace of the headline bench, PMC HBM traffic (+ SQ counters for
# the headline config).  Outputs land in gpurun_out/<tag>_*; copy the summaries into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
b() { name=$1; shift; timeout 400 python bench.py "$@" > $O/${TAG}_bench_$name.json 2> $O/${TAG}_bench_$name.err; python - <<PY
import json
try:
    j = json.load(open("$O/${TAG}_bench_$name.json"))
    print("$name: %.0f img/s  %.0f faces/s  steps %d  dominant %s frac %.3f" % (j["images_per_sec"], j["value"], j["steps"], j["roofline"]["kernel_instance"], j["roofline"]["frac"]))
except Exception as e:
    print("$name: FAILED", e)
PY
}
b b8_448_fp16
b int8_0517_b32 --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline
b int8_mnet25_b32 --precision int8 --model mnet25 --batch 32 --no-cpu-baseline
b 1280x896_b1_fp16 --height 896 --width 1280 --batch 1 --no-cpu-baseline
b b32_448_fp16 --batch 32 --no-cpu-baseline --host-seconds 0
b fp32 --precision fp32 --no-cpu-baseline --host-seconds 0
cp $O/bench_kernels.json $O/${TAG}_bench_kernels_hip_events_fp32.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --host-seconds 0 > /dev/null 2>&1; cp $O/bench_kernels.json $O/${TAG}_bench_kernels_hip_events.json
# kernel trace of the headline bench (no counters in this run)
cd /tmp && export TMPDIR=/tmp
for cfg in "b8_448_fp16:" "int8_mnet25_b32:--precision int8 --model mnet25 --batch 32" "1280x896_b1_fp16:--height 896 --width 1280 --batch 1"; do
    name=${cfg%%:*}; args=${cfg#*:}
    rm -rf $O/${TAG}_ktrace_$name
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ktrace_$name -o kt --output-format csv -- python $R/bench.py --timed-only --steps 2000 --min-seconds 0.5 $args > $O/${TAG}_ktrace_$name.log 2>&1
    f=$(find $O/${TAG}_ktrace_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" $O/${TAG}_bench_${name}_kernel_trace.csv
    rm -rf $O/${TAG}_ktrace_$name
done
# the same with ONE lane: no other launch shares the chip, so these averages are the ones that agree with the HIP-event figures
rm -rf $O/${TAG}_ktrace_lanes1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ktrace_lanes1 -o kt --output-format csv -- python $R/bench.py --timed-only --lanes 1 --steps 2000 --min-seconds 0.5 > $O/${TAG}_ktrace_lanes1.log 2>&1
f=$(find $O/${TAG}_ktrace_lanes1 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/${TAG}_bench_b8_448_fp16_kernel_trace_lanes1.csv
rm -rf $O/${TAG}_ktrace_lanes1
cd $R
bash tools/profile_round.sh ${TAG}_n256_448_fp16 256 fp16 mnet25 448 448 8 sq 2>&1 | tail -24
bash tools/profile_round.sh ${TAG}_n32_1280x896_fp16 32 fp16 mnet25 896 1280 1 2>&1 | tail -2
bash tools/profile_round.sh ${TAG}_n256_448_int8 256 int8 mnet25 448 448 32 2>&1 | tail -2
ls $O | grep ${TAG}_ | head -40
