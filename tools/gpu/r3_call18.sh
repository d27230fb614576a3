#!/bin/bash
# round 3, GPU call 18: int8 epilogues without the separate v_rndne_f32 (v_cvt_pk_u8_f32 rounds to nearest even itself): int8 tests, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c18
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "int8 or calibration" > $O/pytest_int8.log 2>&1
echo "rc $?" >> $O/pytest_int8.log
for rep in 1 2; do for which in base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_${which}_$rep > $O/kbench_int8_${which}_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --precision int8 --batch 32 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep int8 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done; done
grep -v "compute time" $O/pytest_int8.log | tail -3; grep -h "==\|  stem" $O/kbench_*.txt; cat $O/pipe.log
