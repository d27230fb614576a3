#!/bin/bash
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
