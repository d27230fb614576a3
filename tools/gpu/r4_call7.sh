#!/bin/bash
# round 4, GPU call 7: warp-specialised depthwise/pointwise blocks (dwpw_ws_kernel, 64- and 128-channel stride-1 blocks +- lateral): parity with 2 and 3 halo
# buffers (incl. the bit-exact int8 suite), then A/B per kernel against K_b
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c7
mkdir -p $O
cd $R
for v in 2 3; do
  RF_DWPWWS=$v timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3; do for ws in 0 2 3; do
  RF_DWPWWS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_dws${ws}_$rep > $O/kbench_fp16_dws${ws}_$rep.txt 2>&1
  RF_DWPWWS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_dws${ws}_$rep > $O/kbench_int8_dws${ws}_$rep.txt 2>&1
done; done
for v in 2 3; do grep -v "compute time" $O/pytest_$v.log | tail -3; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw<64,64,s1,lat>\|dwpw<128,128' $f | awk '{printf "%s ", $2}')"; done
