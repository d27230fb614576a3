#!/bin/bash
# round 4, GPU call 8: the whole -m gpu suite on the tree with the ADVICE fixes, the rehearsal tests and the int8 self-check; smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c8
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
grep -v "compute time" $O/pytest.log | tail -15; tail -2 $O/smoke.log
