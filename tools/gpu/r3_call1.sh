#!/bin/bash
# round 3, GPU call 1: the whole -m gpu suite (new: bit-exact int8 parity, fp16 contract over 208 frames) + single-lane kernel traces
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=10 -s -k "int8_engine_is_bit_exact or fp16_contract" > $O/pytest_new.log 2>&1
echo "new tests rc $?" >> $O/pytest_new.log
timeout 600 python -m pytest tests -m gpu -q --durations=10 --deselect tests/test_gpu_parity.py::test_fp16_contract_over_200_frames_both_models_both_sizes -k "not int8_engine_is_bit_exact" > $O/pytest_rest.log 2>&1
echo "rest rc $?" >> $O/pytest_rest.log
cd /tmp; export TMPDIR=/tmp
for cfg in "int8 mnet25 32" "fp16 mnet25 8"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$1 -o t -- python $R/bench.py --precision $1 --model $2 --batch $3 --lanes 1 --timed-only --no-cpu-baseline > $O/trace_$1.log 2>&1
  db=$(find $O/trace_$1 -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $O/trace_$1_lanes1.txt > /dev/null
  rm -rf $O/trace_$1
done
tail -3 $O/pytest_new.log; tail -3 $O/pytest_rest.log
