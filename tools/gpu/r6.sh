#!/bin/bash
# round 6: ONE parametrised script for every GPU call of the round (as tools/gpu/r5.sh).
#   gpurun --timeout T -- 'bash tools/gpu/r6.sh <tag> <recipe> [<recipe> ...]'      output -> gpurun_out/r6_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/r6_$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
for recipe in "$@"; do
  echo "=== $recipe"
  case $recipe in
    suite)
      ( time timeout 1400 python -m pytest tests -m gpu -q --durations=8 -s ) > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
      grep -v "compute time" $O/pytest.log | grep -E "passed|failed|contract|^rc|real|s call" | tail -24 ;;
    int8_tests)
      ( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "int8 or calibration or concurrent" ) > $O/pytest_int8.log 2>&1; echo "rc $?" >> $O/pytest_int8.log
      grep -v "compute time" $O/pytest_int8.log | grep -E "passed|failed|Error|assert|int8|^rc|real" | tail -20 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    calibrate)      # both models: per-channel amax x margin + calibrated weights -> $O/cal/<stem>.{table.int8,qweights.int8,rfw}
      mkdir -p $O/cal
      for m in mnet25 mnet-deconv-0517; do
        timeout 900 python tools/calibrate_int8.py --model $m --per-channel --rule amax --margin ${MARGIN:-1.25} --frames ${CAL_FRAMES:-48} --gptq \
            --out $O/cal/$m.table.int8 --out-rfw $O/cal/$m.rfw > $O/calibrate_$m.log 2>&1; echo "calibrate $m rc $?"; grep -E "weights calibrated|wrote" $O/calibrate_$m.log
      done ;;
    contract_old)   # the int8 contract numbers of the assets as shipped (+ fp16 through the same metric)
      timeout 1200 python tools/probes/int8_contract_probe.py --fp16 --json $O/int8_contract_shipped_assets.json > $O/int8_contract_shipped_assets.txt 2>&1; cat $O/int8_contract_shipped_assets.txt | grep contract ;;
    contract_new)   # ... of the calibration made by `calibrate` in this call
      timeout 1200 python tools/probes/int8_contract_probe.py --assets $O/cal --json $O/int8_contract_new_calibration.json > $O/int8_contract_new_calibration.txt 2>&1; cat $O/int8_contract_new_calibration.txt | grep contract ;;
    kbench_int8)
      timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r6_${TAG}_int8 > $O/kbench_int8.txt 2>&1; cat $O/kbench_int8.txt | cut -c1-70 ;;
    kbench_fp16)
      timeout 200 python tools/kbench.py --n 256 --tag r6_${TAG}_fp16 > $O/kbench_fp16.txt 2>&1; cat $O/kbench_fp16.txt | cut -c1-70 ;;
    bench)
      ( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
      cp gpurun_out/bench_kernels.json $O/bench_kernels.json 2>/dev/null; cp gpurun_out/bench_pipeline_trace.json $O/ 2>/dev/null
      tail -3 $O/bench_time.txt
      python - <<PY
import json
j=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("value", round(j["value"]), "images/s", round(j["images_per_sec"]), "sync_batch ms", round(j["sync_batch"]["ms_per_call"],4))
print("roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if not isinstance(v,(dict,list))})
for c in j.get("configs", []):
    print(" cfg", c["id"], round(c["images_per_sec"]), "img/s  sync", round(c["sync_batch"]["ms_per_call"],4), "ms", c.get("dominant_kernel"), c.get("bound"), c.get("bound_frac"))
PY
      ;;
    bench_driver)
      timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-pmc > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/bench_driver.err
      python -c "import json;j=json.loads(open('$O/bench_driver_invocation_steps20_warmup5.json').read().strip().splitlines()[-1]);print('driver invocation', round(j['images_per_sec']), 'img/s', round(j['value']), 'faces/s')" ;;
    *) echo "unknown recipe $recipe" ;;
  esac
done
