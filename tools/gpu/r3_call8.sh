#!/bin/bash
# round 3, GPU call 8: A/B of a second upload stream for host frames (RF_COPY_STREAMS), interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c8
mkdir -p $O
cd $R
for rep in 1 2; do for cs in 1 2; do RF_COPY_STREAMS=$cs timeout 120 python tools/probes/host_rate.py 1.5 >> $O/host_rate.log 2>&1; done; done
grep images $O/host_rate.log
