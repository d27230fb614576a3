#!/bin/bash
# round 3, GPU call 12: lanes sweep for the int8 engine (its kernels leave more of the chip idle than the fp16 ones: three lanes give +14 % over
# the single-lane kernel sum, against +5 % in fp16), interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c12
mkdir -p $O
cd $R
for rep in 1 2; do for lanes in 3 4 5 6; do
  timeout 200 python bench.py --precision int8 --model mnet25 --batch 32 --lanes $lanes --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('int8 lanes $lanes rep $rep', round(j['images_per_sec']))" >> $O/lanes.log
done; done
cat $O/lanes.log
