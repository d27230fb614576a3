cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "compute time" | tail -4 ) 2>&1
for i in 1 2; do for v in 0 1; do for l in 3 1; do
RF_SNAKE=$v timeout 300 python bench.py --timed-only --lanes $l > gpurun_out/c_b.json 2> gpurun_out/c_b.err
python -c "
import json; j=json.load(open('gpurun_out/c_b.json')); print('snake $v lanes $l img/s %.0f' % (j['images_per_sec']))"
done; done; done
