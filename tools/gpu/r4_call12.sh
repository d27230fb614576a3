#!/bin/bash
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
