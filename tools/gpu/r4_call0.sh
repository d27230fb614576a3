#!/bin/bash
# round 4, GPU call 0: LDS-DMA semantics probe (what the chained depthwise/pointwise kernel's staging will rely on) + per-kernel baselines of HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c0
mkdir -p $O
cd $R
timeout 60 tools/probes/lds_dma.bin > $O/lds_dma.txt 2>&1; echo "rc $?" >> $O/lds_dma.txt
timeout 200 python tools/kbench.py --n 256 --tag r4c0_fp16 > $O/kbench_fp16.txt 2>&1
timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r4c0_int8 > $O/kbench_int8.txt 2>&1
cat $O/lds_dma.txt; grep -h "==" $O/kbench_*.txt
