#!/bin/bash
# round 4, GPU call 30: stem2 V2 bits: 1 = planar conv2 tile (layout only), 2 = conv3 -> conv4 chained in registers, 3 = both: identity of (1), A/B of all
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c30
mkdir -p $O
cd $R
timeout 300 python tools/probes/knob_equal.py --precision 1 RF_STEM2_V2=0 RF_STEM2_V2=2 RF_STEM2_V2=3 > $O/equal_fp16.txt 2>&1
for rep in 1 2 3; do for v in 0 1 2 3; do
  RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_s2v${v}_$rep > $O/kbench_fp16_s2v${v}_$rep.txt 2>&1
done; done
cat $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'stem2' $f | awk '{printf "%s ", $2}')"; done
