#!/bin/bash
# round 4, GPU call 5: warp-specialised SSH conv with 2 or 3 halo buffers (producer 1 or 2 tiles ahead, counted vmcnt) x prefetch depth 2 / 3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c5
mkdir -p $O
cd $R
for v in 22 32 33; do
  RF_CONV3WS=$v timeout 600 python -m pytest tests -m gpu -q -x -k "every_fused_op or bit_exact or determinism or odd_net_size" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3 4; do for ws in 0 22 23 32 33; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_ws${ws}_$rep > $O/kbench_fp16_ws${ws}_$rep.txt 2>&1
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_ws${ws}_$rep > $O/kbench_int8_ws${ws}_$rep.txt 2>&1
done; done
for v in 22 32 33; do grep -v "compute time" $O/pytest_$v.log | tail -2; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') $(grep -h 'conv3x3<64,48' $f | awk '{print $2}')"; done
