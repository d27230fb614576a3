cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RF_STEM2=2 timeout 900 python -m pytest tests -m gpu -q -x -k "fused_op or synthetic or odd_net or fixture" 2>&1 | grep -v "compute time" | tail -3
for v in 1 2 1 2; do
RF_STEM2=$v timeout 200 python tools/kbench.py --n 256 --tag cur 2>&1 | grep -E "stem2" | sed "s/^/v$v /"
done
