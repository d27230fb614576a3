cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c_b1.json 2> gpurun_out/c_b1.err ) 2>&1 | grep real; echo rc $?
python -c "
import json; j=json.loads(open('gpurun_out/c_b1.json').read().strip().splitlines()[-1]); print({k:j[k] for k in ('n_gpus','value','images_per_sec','steps','steps_requested','ms_per_step','timed_seconds')}); print(j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
timeout 300 python bench.py --gpus 2 --oversubscribe --no-cpu-baseline --host-seconds 0 --steps 20 --warmup 5 > gpurun_out/c_b2.json 2> gpurun_out/c_b2.err; echo rc $?
python -c "
import json; j=json.loads(open('gpurun_out/c_b2.json').read().strip().splitlines()[-1]); print({k:j[k] for k in ('n_gpus','value','images_per_sec','steps','ms_per_step','timed_seconds')}); print(j['result_gather'])"
