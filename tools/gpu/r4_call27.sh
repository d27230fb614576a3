#!/bin/bash
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
