#!/bin/bash
# round 3, GPU call 13: fewer barriers per tile (dwpw without lateral: 3 -> 2; conv3x3 double buffered: 2 -> 1; ssh_tail: 3 -> 2): whole suite
# (twice for the determinism / batch-invariance tests: a missing barrier is a race), then A/B against the previous build, fp16 + int8, per kernel
# and three-lane pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c13
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 600 python -m pytest tests -m gpu -q -x -k "determinism or batch_composition or bit_exact or synthetic_batch8 or every_fused_op" > $O/pytest_again.log 2>&1
echo "rc $?" >> $O/pytest_again.log
for rep in 1 2; do for which in base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --tag fp16_${which}_$rep > $O/kbench_fp16_${which}_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$lib timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_${which}_$rep > $O/kbench_int8_${which}_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep fp16 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --precision int8 --batch 32 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep int8 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done; done
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -4; tail -2 $O/pytest_again.log; grep -h "==" $O/kbench_*.txt; cat $O/pipe.log
