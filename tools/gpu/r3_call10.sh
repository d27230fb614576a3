#!/bin/bash
# round 3, GPU call 10: co-scheduling experiment -- stem2 capped at 7 / 6 workgroups per CU (RF_STEM2_PAD = 3 / 7 KB of unused LDS) so that another
# lane's memory-bound kernels can run beside it; three-lane pipeline throughput, interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c10
mkdir -p $O
cd $R
for rep in 1 2; do for pad in 0 3 7; do
  RF_STEM2_PAD=$pad timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pad $pad rep $rep lanes3', round(j['images_per_sec']), [round(x) for x in j['regions']['faces_per_sec']])" >> $O/pad.log
done; done
for pad in 0 7; do RF_STEM2_PAD=$pad timeout 200 python bench.py --timed-only --no-cpu-baseline --lanes 4 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pad $pad lanes4', round(j['images_per_sec']))" >> $O/pad.log; done
cat $O/pad.log
