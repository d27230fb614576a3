#!/bin/bash
# round 4, GPU call 29: stem2 V2 (planar conv2 tile: 4-way -> 2-way epilogue writes; conv3 -> conv4 chained in registers: one tile and one barrier less): parity subset + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c29
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or determinism or odd_net_size or fixture_image or contract" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2 3; do for v in 0 1; do
  RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_s2v${v}_$rep > $O/kbench_fp16_s2v${v}_$rep.txt 2>&1
done; done
grep -v "compute time" $O/pytest.log | tail -5
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'stem2' $f | awk '{printf "%s ", $2}')"; done
