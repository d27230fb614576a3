#!/bin/bash
# round 3, GPU call 6: conv0's 1024 offset folded into the bias (u8x4_to_f16 without the subtraction): suite, then per-kernel A/B against the
# previous build (retinaface_amd/lib_base) inside one call, fp16 and int8, twice each (box drift)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do
  for which in base new; do
    lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
    RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --tag fp16_${which}_$rep > $O/kbench_fp16_${which}_$rep.txt 2>&1
    RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_${which}_$rep > $O/kbench_int8_${which}_$rep.txt 2>&1
  done
done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract|int8 front" | tail -6; grep -h "==\|  stem" $O/kbench_*.txt
