#!/bin/bash
# round 4, GPU call 4: warp-specialised SSH conv with the pinned pipeline at prefetch depth 2 / 3 / 4 vs the lock-step kernel (pinned too), 3 repetitions, fp16 + int8
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "every_fused_op or bit_exact or determinism" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for ws in 0 2 3 4; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_ws${ws}_$rep > $O/kbench_fp16_ws${ws}_$rep.txt 2>&1
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_ws${ws}_$rep > $O/kbench_int8_ws${ws}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') $(grep -h 'conv3x3<64,48' $f | awk '{print $2}') $(grep -h 'ssh_tail' $f | awk '{print $2}')"; done
