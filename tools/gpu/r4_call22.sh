#!/bin/bash
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
