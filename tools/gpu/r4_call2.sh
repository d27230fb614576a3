#!/bin/bash
# round 4, GPU call 2: warp-specialised SSH conv (conv3x3_ws_kernel): parity subset (per-op blobs, goldens, bit-exact int8, determinism, 1280x896, odd sizes),
# then A/B against RF_CONV3WS=0 inside the same call, fp16 + int8, per kernel and three-lane pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or fixture_image or bit_exact or determinism or batch_composition or odd_net_size or edge_cases or knob" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do for ws in 0 1; do
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --tag fp16_ws${ws}_$rep > $O/kbench_fp16_ws${ws}_$rep.txt 2>&1
  RF_CONV3WS=$ws timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_ws${ws}_$rep > $O/kbench_int8_ws${ws}_$rep.txt 2>&1
done; done
for ws in 0 1; do
  RF_CONV3WS=$ws timeout 200 python bench.py --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ws $ws fp16 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
  RF_CONV3WS=$ws timeout 200 python bench.py --precision int8 --batch 32 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ws $ws int8 three lanes', round(j['images_per_sec']))" >> $O/pipe.log
done
grep -v "compute time" $O/pytest.log | tail -5; grep -h "==\|conv3x3<64,48" $O/kbench_*.txt; cat $O/pipe.log
