#!/bin/bash
# round 3, GPU call 4: the fused SSH tail (ssh_tail_kernel): whole suite, then per-kernel A/B (RF_SSHTAIL = 0 two launches / 1 fused,
# 4 workgroups per CU / 2 fused, 3 per CU) in fp16 and int8 at 256 images per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c4
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for v in 0 1 2; do
  RF_SSHTAIL=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_sshtail$v > $O/kbench_fp16_sshtail$v.txt 2>&1
  RF_SSHTAIL=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_sshtail$v > $O/kbench_int8_sshtail$v.txt 2>&1
done
grep -v "compute time" $O/pytest.log | tail -6; grep -h "==\|ssh_tail\|conv3x3<16" $O/kbench_*.txt
