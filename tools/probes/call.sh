cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "compute time" | tail -5 ) 2>&1
timeout 600 python tools/probes/iou_margin.py 6 2>&1 | grep -v "compute time\|amdgpu.ids" | tail -4
timeout 200 python tools/kbench.py --n 256 --tag cur 2>&1 | grep -E "^==|stem2"
for i in 1 2; do for v in lib_prev lib; do
RETINAFACE_AMD_LIB=$PWD/retinaface_amd/$v/libretinaface_amd.so timeout 300 python bench.py --timed-only > gpurun_out/c_b.json 2> gpurun_out/c_b.err
python -c "
import json; j=json.load(open('gpurun_out/c_b.json')); print('$v img/s %.0f ms/step %.4f steps %d' % (j['images_per_sec'], j['ms_per_step'], j['steps']))"
done; done
