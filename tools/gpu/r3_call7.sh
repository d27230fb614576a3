#!/bin/bash
# round 3, GPU call 7: whole suite on the offset-folded conv0 build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c7
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=3 -s > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract|int8 front|calibration tool" | tail -8
