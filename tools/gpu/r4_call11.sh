#!/bin/bash
# round 4, GPU call 11: why is the warp-specialised aggregation conv slow?  phase stamps of consumer wave 0 and the producer wave; dwpw2 ring A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c11
mkdir -p $O
cd $R
RF_CONV3UPWS=3 timeout 200 python tools/probes/ws_trace.py 256 > $O/ws_trace_nbuf3.txt 2>&1
RF_CONV3UPWS=2 timeout 200 python tools/probes/ws_trace.py 256 > $O/ws_trace_nbuf2.txt 2>&1
for rep in 1 2 3; do for r in 0 1; do
  RF_DWPW2_RING=$r timeout 200 python tools/kbench.py --n 256 --tag fp16_ring${r}_$rep > $O/kbench_fp16_ring${r}_$rep.txt 2>&1
done; done
RF_DWPW2_RING=1 timeout 600 python -m pytest tests -m gpu -q -x -k "every_fused_op or golden or determinism" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
cat $O/ws_trace_nbuf3.txt $O/ws_trace_nbuf2.txt | grep -v amdgpu.ids
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -h 'dwpw2' $f | awk '{printf "%s ", $2}')"; done; tail -3 $O/pytest.log
