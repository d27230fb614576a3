#!/bin/bash
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
