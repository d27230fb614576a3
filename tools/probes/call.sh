cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -x -k "fused_op or synthetic or odd_net or fixture or edge or unaligned or determinism" 2>&1 | grep -v "compute time" | tail -3 ) 2>&1
for v in lib_prev lib lib_prev lib; do
RETINAFACE_AMD_LIB=$PWD/retinaface_amd/$v/libretinaface_amd.so timeout 200 python tools/kbench.py --n 256 --tag cur 2>&1 | grep -E "stem2" | sed "s/^/$v /"
done
