cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do for v in lib_prev lib; do
RETINAFACE_AMD_LIB=$PWD/retinaface_amd/$v/libretinaface_amd.so timeout 300 python bench.py --timed-only > gpurun_out/c_b.json 2> gpurun_out/c_b.err
python -c "
import json; j=json.load(open('gpurun_out/c_b.json')); print('$v img/s %.0f ms/step %.4f steps %d' % (j['images_per_sec'], j['ms_per_step'], j['steps']))"
done; done
