#!/bin/bash
# round 3, GPU call 2: suite after the multi-GPU scatter / lazy lanes / plan-cache / stem2 DC-centring changes; A/B of the centring on
# the 208-frame fp16 contract; the reworked bench line (3 regions, physical roofline, sync_batch)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=8 -x --deselect tests/test_gpu_parity.py::test_fp16_contract_over_200_frames_both_models_both_sizes > $O/pytest.log 2>&1
echo "suite rc $?" >> $O/pytest.log
for dc in 1 0; do
  RF_STEM2_DC=$dc timeout 600 python -m pytest tests -m gpu -q -s -k fp16_contract 2>&1 | grep -E "fp16 contract|passed|failed" > $O/contract_dc$dc.log
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $?" >> $O/bench_default.err
cp gpurun_out/bench_kernels.json $O/bench_kernels_fp16.json 2>/dev/null
tail -4 $O/pytest.log; cat $O/contract_dc*.log; head -c 1500 $O/bench_default.json
