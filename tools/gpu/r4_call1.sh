#!/bin/bash
# round 4, GPU call 1: phase timelines (s_memtime stamps, probe build) of the kernels the verdict names as latency bound, at 256 images per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c1
mkdir -p $O
cd $R
timeout 120 python tools/probes/phase_trace.py conv3 256 1072 25152 6336 > $O/trace_conv3.txt 2>&1
timeout 120 python tools/probes/phase_trace.py dwpw 256 7168 25088 > $O/trace_dwpw.txt 2>&1
timeout 120 python tools/probes/phase_trace.py dwpw2 256 0 > $O/trace_dwpw2.txt 2>&1
timeout 120 python tools/probes/phase_trace.py stem2 256 0 > $O/trace_stem2.txt 2>&1
tail -n 30 $O/trace_conv3.txt $O/trace_dwpw.txt $O/trace_dwpw2.txt $O/trace_stem2.txt
