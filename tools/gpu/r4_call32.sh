#!/bin/bash
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
