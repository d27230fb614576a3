#!/bin/bash
# round 4, GPU call 35: stem2 RF_STEM2_V2 = 5 (default) vs 7 (+ conv3 -> conv4 chained in registers): what the chain would add on top of the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c35
mkdir -p $O
cd $R
for rep in 1 2 3; do for v in 5 7; do
  RF_STEM2_V2=$v timeout 100 python tools/kbench.py --n 256 --tag fp16_s2v${v}_$rep > $O/kbench_fp16_s2v${v}_$rep.txt 2>&1
done; done
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'stem2' $f | awk '{printf "%s ", $2}')"; done
