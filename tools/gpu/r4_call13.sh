#!/bin/bash
# round 4, GPU call 13: 8x8 tiles for the 256-channel block (the streamed weight matrix is read once per 64 pixels instead of 32): parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c13
mkdir -p $O
cd $R
RF_TILE256=1 timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  RF_TILE256=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_t256_${v}_$rep > $O/kbench_fp16_t256_${v}_$rep.txt 2>&1
  RF_TILE256=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_t256_${v}_$rep > $O/kbench_int8_t256_${v}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest.log | tail -3
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw<256,256' $f | awk '{printf "%s ", $2}')"; done
