#!/bin/bash
# round 4, GPU call 9: warp-specialised aggregation convs (conv3x3_up_ws_kernel: LDS-DMA of lateral halo + coarse patch, blend LDS -> LDS, 8x8 tiles):
# parity with 2 and 3 ring buffers (incl. bit-exact int8 and the integer-blend test), then A/B per kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c9
mkdir -p $O
cd $R
for v in 2 3; do
  RF_CONV3UPWS=$v timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size or fixture_image or integer_blend or edge_cases" > $O/pytest_$v.log 2>&1
  echo "rc $?" >> $O/pytest_$v.log
done
for rep in 1 2 3; do for ws in 0 2 3; do
  RF_CONV3UPWS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_up${ws}_$rep > $O/kbench_fp16_up${ws}_$rep.txt 2>&1
  RF_CONV3UPWS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_up${ws}_$rep > $O/kbench_int8_up${ws}_$rep.txt 2>&1
done; done
for v in 2 3; do grep -v "compute time" $O/pytest_$v.log | tail -3; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'conv3x3<64,64' $f | awk '{printf "%s ", $2}')"; done
