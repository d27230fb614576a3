#!/bin/bash
# round 4, GPU call 14: warp-specialised SSH conv with the 2 + 2 + 1 role split of the GEMM waves (LDS B-fragment traffic 216 -> 144 KB per tile)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c14
mkdir -p $O
cd $R
for v in 132; do
  RF_CONV3WS=$v timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3; do for ws in 1 132 122; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_ws${ws}_$rep > $O/kbench_fp16_ws${ws}_$rep.txt 2>&1
done; for ws in 0 132 122; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_ws${ws}_$rep > $O/kbench_int8_ws${ws}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest_132.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'conv3x3<64,48' $f | awk '{printf "%s ", $2}')"; done
