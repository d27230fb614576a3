#!/bin/bash
# round 4, GPU call 10: producer joins the mid-interval barrier before its issue work (UPADD WS conv), DMA after the first barrier (dwpw WS): A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c10
mkdir -p $O
cd $R
RF_CONV3UPWS=3 RF_DWPWWS=3 timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or integer_blend" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do for ws in 0 2 3; do
  RF_CONV3UPWS=$ws RF_DWPWWS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_v${ws}_$rep > $O/kbench_fp16_v${ws}_$rep.txt 2>&1
  RF_CONV3UPWS=$ws RF_DWPWWS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_v${ws}_$rep > $O/kbench_int8_v${ws}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'conv3x3<64,64\|dwpw<64,64,s1,lat>\|dwpw<128,128' $f | awk '{printf "%s ", $2}')"; done
