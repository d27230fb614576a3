#!/bin/bash
# round 3, GPU call 11: dwpw2 at 4 workgroups per CU (128 VGPRs: 3 + 2 unit split, biases read from LDS): fp16 tests, then A/B against the previous
# build (per-kernel and three-lane pipeline), interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c11
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "not int8 and not calibration" -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do for which in base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --tag fp16_${which}_$rep > $O/kbench_fp16_${which}_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done; done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; grep -h "==\|dwpw2" $O/kbench_*.txt; cat $O/pipe.log
