#!/bin/bash
# round 4, GPU call 6: the tightened fp16 bands / order / per-layer bars on the GPU, and the new default bench line (configs array, physical frac)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -s -k "every_fused_op or golden or fixture_image or fp16_contract" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
grep -v "compute time" $O/pytest.log | grep -E "fp16 layer|fp16 contract|passed|failed|Error|assert" | tail -40
tail -3 $O/bench_time.txt; tail -5 $O/bench_default.err
python - <<'P'
import json
j=json.loads(open('/root/repo/gpurun_out/r4c6/bench_default.json').read().strip().splitlines()[-1])
print(round(j['images_per_sec']), j['roofline']['bound'], j['roofline']['frac'], j['roofline']['unit'])
for c in j.get('configs', []): print(c['id'], round(c['images_per_sec']), round(c['faces_per_sec']), c['dominant_kernel'], c.get('bound'), c.get('bound_frac'), c.get('hbm_frac_measured'))
print(j.get('cpu_baseline',{}).get('value'))
P
