#!/bin/bash
# round 4, GPU call 21: K_b'' -- the 64- / 128-channel depthwise-pointwise blocks with the halo DMA and the stores spread over the four GEMM waves
# (RF_DWPWWS=12 / 13: 2 / 3 halo buffers): identity + A/B against K_b
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c21
mkdir -p $O
cd $R
timeout 600 python tools/probes/knob_equal.py --precision 2 RF_DWPWWS=12 RF_DWPWWS=13 > $O/equal_int8.txt 2>&1
timeout 600 python tools/probes/knob_equal.py --precision 1 RF_DWPWWS=12 RF_DWPWWS=13 > $O/equal_fp16.txt 2>&1
for rep in 1 2; do for v in 0 12 13; do
  RF_DWPWWS=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_dd${v}_$rep > $O/kbench_fp16_dd${v}_$rep.txt 2>&1
  RF_DWPWWS=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_dd${v}_$rep > $O/kbench_int8_dd${v}_$rep.txt 2>&1
done; done
cat $O/equal_int8.txt $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw<64,64\|dwpw<128,128' $f | awk '{printf "%s ", $2}')"; done
