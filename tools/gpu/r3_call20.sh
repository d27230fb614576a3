#!/bin/bash
# round 3, GPU call 20: int8 stem with 1 / scale folded into its pointwise weights (no fma in the epilogue): int8 tests, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c20
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -s -k "int8 or calibration" > $O/pytest_int8.log 2>&1
echo "rc $?" >> $O/pytest_int8.log
for which in base new base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_$which 2>&1 | grep "==\|  stem" >> $O/kbench.log
done
grep -v "compute time" $O/pytest_int8.log | grep -E "passed|failed|front end|Error" | tail -5; cat $O/kbench.log
