#!/bin/bash
# round 4, GPU call 16: tile shapes of the int8 64- and 128-channel blocks (8x8, 4x16, 8x16 instead of 4x8): bit-identity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c16
mkdir -p $O
cd $R
timeout 900 python tools/probes/knob_equal.py --precision 2 RF_TILE128=1 RF_TILE128=2 RF_TILE128=3 RF_TILE64=1 RF_TILE64=2 > $O/equal_int8.txt 2>&1
timeout 600 python tools/probes/knob_equal.py --precision 1 RF_TILE128=1 RF_TILE128=2 RF_TILE64=1 > $O/equal_fp16.txt 2>&1
for rep in 1 2; do
  for v in 0 1 2 3; do RF_TILE128=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_t128_${v}_$rep > $O/kbench_int8_t128_${v}_$rep.txt 2>&1; done
  for v in 1 2; do RF_TILE64=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_t64_${v}_$rep > $O/kbench_int8_t64_${v}_$rep.txt 2>&1; done
  for v in 0 1 2; do RF_TILE128=$v timeout 200 python tools/kbench.py --n 256 --tag fp16_t128_${v}_$rep > $O/kbench_fp16_t128_${v}_$rep.txt 2>&1; done
  RF_TILE64=1 timeout 200 python tools/kbench.py --n 256 --tag fp16_t64_1_$rep > $O/kbench_fp16_t64_1_$rep.txt 2>&1
done
cat $O/equal_int8.txt $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw<64,64\|dwpw<128,128' $f | awk '{printf "%s ", $2}')"; done
