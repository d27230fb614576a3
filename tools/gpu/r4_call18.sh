#!/bin/bash
# round 4, GPU call 18: larger tiles for the int8 engine's big-map blocks (16->32 s2, 32->32, 32->64 s2, 64->128 s2): bit-identity + A/B; library now built in VGPR-form
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c18
mkdir -p $O
cd $R
timeout 900 python tools/probes/knob_equal.py --precision 2 RF_TILE_A=1 RF_TILE_A=2 RF_TILE_B=1 RF_TILE_B=2 RF_TILE_C=1 RF_TILE_C=2 RF_TILE_D=1 RF_TILE128=2 > $O/equal_int8.txt 2>&1
for rep in 1 2; do
  timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_base_$rep > $O/kbench_int8_base_$rep.txt 2>&1
  for k in RF_TILE_A=1 RF_TILE_A=2 RF_TILE_B=1 RF_TILE_B=2 RF_TILE_C=1 RF_TILE_C=2 RF_TILE_D=1 RF_TILE128=2; do
    env $k timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_${k/=/_}_$rep > $O/kbench_int8_${k/=/_}_$rep.txt 2>&1
  done
done
cat $O/equal_int8.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -v '==' $f | grep us | awk '{printf "%s ", $2}')"; done
