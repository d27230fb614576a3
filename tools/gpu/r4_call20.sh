#!/bin/bash
# round 4, GPU call 20: one synchronous call of 8 / 1 device-resident frames: hipGraphLaunch vs eager launches (host trace of both)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c20
mkdir -p $O
cd $R
for b in 8 1; do for g in 1 0; do
  RF_HOST_TRACE=1 timeout 200 python tools/probes/sync_latency.py $b $g > $O/sync_b${b}_g${g}.txt 2>&1
done; done
tail -n 12 $O/sync_*.txt
