#!/bin/bash
# round 3, GPU call 14: final tree -- whole suite, smoke(), three-lane pipeline A/B against the build before the barrier changes (fp16, int8),
# the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c14
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
for rep in 1 2; do for which in base new; do
  lib=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && lib=$R/retinaface_amd/lib_base/libretinaface_amd.so
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep fp16 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
  RETINAFACE_AMD_LIB=$lib timeout 200 python bench.py --precision int8 --batch 32 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$which rep $rep int8 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done; done
timeout 400 python bench.py > $O/bench_b8_448_fp16.json 2> $O/e0.err; cp gpurun_out/bench_kernels.json $O/kernels_b8_448_fp16.json
timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline > $O/bench_int8_mnet25_b32.json 2> $O/e2.err; cp gpurun_out/bench_kernels.json $O/kernels_int8_mnet25_b32.json
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract" | tail -3; tail -2 $O/smoke.log; cat $O/pipe.log
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], round(j['images_per_sec']), round(j['value']))"; done
