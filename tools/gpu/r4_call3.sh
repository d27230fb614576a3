#!/bin/bash
# round 4, GPU call 3: B-fragment software pipeline pinned with sched_barrier in gemm_stationary (the compiler had sunk every ds_read to its MFMA):
# parity subset, then A/B against the round-3 library (lib_base), fp16 + int8, WS conv on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or bit_exact or determinism or odd_net_size" > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
for rep in 1 2; do
  RETINAFACE_AMD_LIB=$R/retinaface_amd/lib_base/libretinaface_amd.so timeout 200 python tools/kbench.py --n 256 --tag fp16_base_$rep > $O/kbench_fp16_base_$rep.txt 2>&1
  RF_CONV3WS=0 timeout 200 python tools/kbench.py --n 256 --tag fp16_pin_ws0_$rep > $O/kbench_fp16_pin_ws0_$rep.txt 2>&1
  RF_CONV3WS=1 timeout 200 python tools/kbench.py --n 256 --tag fp16_pin_ws1_$rep > $O/kbench_fp16_pin_ws1_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$R/retinaface_amd/lib_base/libretinaface_amd.so timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_base_$rep > $O/kbench_int8_base_$rep.txt 2>&1
  RF_CONV3WS=0 timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_pin_ws0_$rep > $O/kbench_int8_pin_ws0_$rep.txt 2>&1
  RF_CONV3WS=1 timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_pin_ws1_$rep > $O/kbench_int8_pin_ws1_$rep.txt 2>&1
done
grep -v "compute time" $O/pytest.log | tail -3; grep -h "==" $O/kbench_*.txt
