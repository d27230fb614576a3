cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "3 32" "3 48" "3 64" "2 64" "3 32" "3 64"; do set -- $cfg
timeout 300 python bench.py --timed-only --lanes $1 --coalesce $2 > gpurun_out/c_b.json 2> gpurun_out/c_b.err
python -c "
import json; j=json.load(open('gpurun_out/c_b.json')); print('lanes $1 coalesce $2: img/s %.0f ms/step %.4f steps %d' % (j['images_per_sec'], j['ms_per_step'], j['steps']))" || tail -3 gpurun_out/c_b.err
done
