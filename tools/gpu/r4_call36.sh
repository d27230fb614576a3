#!/bin/bash
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
