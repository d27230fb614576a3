#!/bin/bash
# round 5: ONE parametrised script for every GPU call of the round (rounds 3-4 committed one near-identical script per call).
#   gpurun --timeout T -- 'bash tools/gpu/r5.sh <tag> <recipe> [<recipe> ...]'      output -> gpurun_out/r5_<tag>/
# recipes: suite | smoke | bench | bench_driver | kbench_fp16 | kbench_int8 | kbench_ab (product vs lib_base) | sync_gaps | libleg | probes | trace1 | bench2
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/r5_$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
for recipe in "$@"; do
  echo "=== $recipe"
  case $recipe in
    suite)
      ( time timeout 1400 python -m pytest tests -m gpu -q --durations=5 -s ) > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
      grep -v "compute time" $O/pytest.log | grep -E "passed|failed|fp16 contract|^rc|real" | tail -8 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    bench)
      ( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
      cp gpurun_out/bench_kernels.json $O/bench_kernels.json 2>/dev/null; cp gpurun_out/bench_pipeline_trace.json $O/ 2>/dev/null
      tail -3 $O/bench_time.txt
      python - <<PY
import json
j=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("value", round(j["value"]), "images/s", round(j["images_per_sec"]), "sync_batch ms", round(j["sync_batch"]["ms_per_call"],4), "burst ms", round(j["burst"]["ms"],3))
print("roofline", r["bound"], r["frac"], "kernel_ms", round(r["kernel_ms"],4), "in pipeline", r.get("kernel_ms_in_pipeline"), "useful", {k:(round(v,4) if isinstance(v,float) else v) for k,v in r["useful"].items() if k.endswith("frac")})
print("whole path ms", round(r["whole_path"]["kernels_ms_per_launch_sequence"],4), "in pipeline", r["whole_path"].get("kernels_ms_in_pipeline"))
for c in j.get("configs", []):
    print(" cfg", c["id"], round(c["images_per_sec"]), "img/s  sync", round(c["sync_batch"]["ms_per_call"],4), "ms", c.get("dominant_kernel"), c.get("bound"), c.get("bound_frac"))
print("cpu", j.get("cpu_baseline",{}).get("images_per_sec"), j.get("cpu_baseline",{}).get("gpu_faces_identical_to_oracle"), "host", {k:round(v["images_per_sec"]) for k,v in j.get("host_frames",{}).items() if isinstance(v,dict) and "images_per_sec" in v})
PY
      ;;
    bench_driver)
      timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-pmc > $O/bench_driver_invocation_steps20_warmup5.json 2> $O/bench_driver.err
      python -c "import json;j=json.loads(open('$O/bench_driver_invocation_steps20_warmup5.json').read().strip().splitlines()[-1]);print('driver invocation', round(j['images_per_sec']), 'img/s', round(j['value']), 'faces/s')" ;;
    kbench_fp16)
      timeout 200 python tools/kbench.py --n 256 --tag r5_${TAG}_fp16 > $O/kbench_fp16.txt 2>&1; cat $O/kbench_fp16.txt | cut -c1-70 ;;
    kbench_int8)
      timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r5_${TAG}_int8 > $O/kbench_int8.txt 2>&1; cat $O/kbench_int8.txt | cut -c1-70 ;;
    kbench_ab)      # A/B inside one call through the PROBE build: $AB_KNOB=$AB_A vs $AB_KNOB=$AB_B (default: stem2 without / with the conv3 -> conv4 register chain), interleaved
      K=${AB_KNOB:-RF_STEM2_V2}; A=${AB_A:-5}; B=${AB_B:-7}; P=${AB_PREC:-fp16}; BT=${AB_BATCH:-8}
      for rep in 1 2 3; do for v in $A $B; do
        env RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so $K=$v timeout 100 python tools/kbench.py --n 256 --precision $P --batch $BT --tag ${K}_${v}_$rep > $O/kbench_${K}_${v}_$rep.txt 2>&1
      done; done
      for f in $O/kbench_${K}_*.txt; do echo "$(basename $f) total $(grep -h '==' $f | awk '{print $8}') | $(grep -h -E "${AB_GREP:-stem2}" $f | awk '{printf "%s %s  ", $1, $2}')"; done ;;
    sync_gaps)
      for cfg in "8 448 448 fp16" "1 896 1280 fp16" "32 448 448 int8"; do
        t=$(echo $cfg | tr ' ' '_')
        rm -rf /tmp/kt_$t; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d /tmp/kt_$t -o kt -- python $R/tools/probes/sync_gaps.py run $cfg > $O/sync_gaps_run_$t.log 2>&1 )
        db=$(find /tmp/kt_$t -name "*.db" | head -1)
        python tools/probes/sync_gaps.py report $db > $O/sync_gaps_$t.txt 2>&1; head -3 $O/sync_gaps_$t.txt
      done ;;
    libleg)
      timeout 300 python bench.py --library-devices 8 > $O/bench_library_leg_8_engines_one_gpu.json 2> $O/libleg.err
      python -c "import json;j=json.loads(open('$O/bench_library_leg_8_engines_one_gpu.json').read().strip().splitlines()[-1])['library_multi_device'];print({k:j[k] for k in ('devices','ms_per_call','images_per_sec','frames_scattered_per_call','detections_identical_to_single_engine','forced_scatter_rehearsal')}, j['single_engine_same_call'])" ;;
    bench2)
      timeout 400 python bench.py --gpus 2 --oversubscribe --no-cpu-baseline --host-seconds 0 > $O/bench_2ranks_one_gpu.json 2> $O/bench2.err
      python -c "import json;j=json.loads(open('$O/bench_2ranks_one_gpu.json').read().strip().splitlines()[-1]);print('2 ranks one gpu', j['n_gpus'], round(j['images_per_sec']), j.get('library_multi_device',{}).get('images_per_sec'), j.get('library_multi_device',{}).get('error'))" ;;
    probes)
      ( time RF_PROBE_TESTS=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k probe_knob ) > $O/pytest_probe_knobs.log 2>&1; tail -4 $O/pytest_probe_knobs.log ;;
    trace1)
      rm -rf /tmp/kt1; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt -- python $R/bench.py --timed-only --lanes 1 --no-pmc > $O/trace1_bench.json 2> $O/trace1.err )
      db=$(find /tmp/kt1 -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_fp16.txt > /dev/null; head -24 $O/kernel_trace_lanes1_fp16.txt | cut -c1-60,104-190 ;;
    floor)
      ( cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 launch_floor.cpp -o launch_floor.bin 2>/dev/null; timeout 120 ./launch_floor.bin ) > $O/launch_floor.txt 2>&1; cat $O/launch_floor.txt ;;
    cumask)
      ( cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 cu_mask.cpp -o cu_mask.bin 2>/dev/null; timeout 120 ./cu_mask.bin ) > $O/cu_mask.txt 2>&1; cat $O/cu_mask.txt ;;
    cusplit)        # lanes confined to complementary halves of every XCD's CUs (RF_CU_SPLIT=1, probe build) vs the whole chip per lane, interleaved
      for rep in 1 2; do for cfg in "0 3" "1 2" "1 4" "0 2" "1 6"; do set -- $cfg
        env RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_CU_SPLIT=$1 timeout 200 python bench.py --lanes $2 --no-cpu-baseline --no-extra-configs --no-pmc --no-pipeline-trace --host-seconds 0 --regions 2 > $O/bench_split$1_lanes$2_$rep.json 2> $O/bench_split$1_lanes$2_$rep.err
        python -c "import json;j=json.loads(open('$O/bench_split$1_lanes$2_$rep.json').read().strip().splitlines()[-1]);print('split $1 lanes $2 rep $rep:', round(j['images_per_sec']), 'img/s  burst', round(j['burst']['ms'],3), 'ms  sync', round(j['sync_batch']['ms_per_call'],4))" 2>&1 | tail -1
      done; done ;;
    wide_i8)        # int8 engine on K_b(8): bit-identity of the variants, then the A/B
      RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so timeout 300 python tools/probes/knob_equal.py --precision 2 --model mnet25 --n 32 RF_WIDE_I8=1 RF_WIDE_I8=2 > $O/knob_equal_wide_i8.txt 2>&1; cat $O/knob_equal_wide_i8.txt
      RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so timeout 300 python tools/probes/knob_equal.py --precision 1 --model mnet25 --n 16 RF_WIDE256=0 RF_WIDE128=0 RF_WIDE128=2 > $O/knob_equal_wide_fp16.txt 2>&1; cat $O/knob_equal_wide_fp16.txt
      for rep in 1 2 3; do for v in 0 1 2; do
        env RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_WIDE_I8=$v timeout 100 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag wide_i8_${v}_$rep > $O/kbench_wide_i8_${v}_$rep.txt 2>&1
      done; done
      for f in $O/kbench_wide_i8_*.txt; do echo "$(basename $f) total $(grep -h '==' $f | awk '{print $8}') | $(grep -h -E "dwpw<128,128|dwpw<256" $f | awk '{printf "%s %s  ", $1, $2}')"; done ;;
    kbench_libs)    # the product library of this tree vs the previous build kept in retinaface_amd/lib_base (git-ignored, travels with the snapshot), interleaved
      P=${AB_PREC:-fp16}; BT=${AB_BATCH:-8}
      for rep in 1 2 3; do
        RETINAFACE_AMD_LIB=$R/retinaface_amd/lib_base/libretinaface_amd.so timeout 100 python tools/kbench.py --n 256 --precision $P --batch $BT --tag base_${P}_$rep > $O/kbench_base_${P}_$rep.txt 2>&1
        timeout 100 python tools/kbench.py --n 256 --precision $P --batch $BT --tag new_${P}_$rep > $O/kbench_new_${P}_$rep.txt 2>&1
      done
      for f in $O/kbench_base_${P}_*.txt $O/kbench_new_${P}_*.txt; do echo "$(basename $f) total $(grep -h '==' $f | awk '{print $8}') | $(grep -h -E "${AB_GREP:-head}" $f | awk '{printf "%s %s  ", $1, $2}')"; done ;;
    bench_libs)     # pipeline throughput (the metric point, three lanes) of this tree's product library vs retinaface_amd/lib_base, interleaved
      for rep in 1 2 3; do for which in base new; do
        L=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which = base ] && L=$R/retinaface_amd/lib_base/libretinaface_amd.so
        RETINAFACE_AMD_LIB=$L timeout 200 python bench.py ${BENCH_ARGS} --no-cpu-baseline --no-extra-configs --no-pmc --no-pipeline-trace --host-seconds 0 --regions 2 > $O/bench_${which}_$rep.json 2> $O/bench_${which}_$rep.err
        python -c "import json;j=json.loads(open('$O/bench_${which}_$rep.json').read().strip().splitlines()[-1]);print('$which rep $rep:', round(j['images_per_sec']), 'img/s  burst', round(j['burst']['ms'],3), 'ms  sync', round(j['sync_batch']['ms_per_call'],4), ' dominant', j['roofline']['kernel_instance'], round(j['roofline']['kernel_ms']*1e3,1), 'us  sum', round(j['roofline']['whole_path']['kernels_ms_per_launch_sequence']*1e3,1))" 2>&1 | tail -1
      done; done ;;
    trace_int8)
      rm -rf /tmp/kt8; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt8 -o kt -- python $R/bench.py --timed-only --lanes 1 --no-pmc --precision int8 --batch 32 > $O/trace_int8_bench.json 2> $O/trace_int8.err )
      db=$(find /tmp/kt8 -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_int8.txt > /dev/null; head -24 $O/kernel_trace_lanes1_int8.txt | cut -c1-60,104-190 ;;
    bench_int8)
      timeout 600 python bench.py --precision int8 --batch 32 --no-cpu-baseline --host-seconds 0 > $O/bench_int8_mnet25_b32.json 2> $O/bench_int8.err; cp gpurun_out/bench_kernels.json $O/bench_kernels_int8.json 2>/dev/null
      python -c "import json;j=json.loads(open('$O/bench_int8_mnet25_b32.json').read().strip().splitlines()[-1]);r=j['roofline'];print('int8 mnet25 b32', round(j['images_per_sec']), 'img/s', r['bound'], r['frac'], 'sum', r['whole_path']['kernels_ms_per_launch_sequence'], 'in pipeline', r['whole_path'].get('kernels_ms_in_pipeline'))" ;;
    hostcopy)       # host-frame pipeline (rf_enqueue_batch from pageable / registered caller memory): staging copy with memcpy (RF_NT_COPY=0) vs non-temporal stores (1), probe build, interleaved
      for rep in 1 2 3; do for v in 0 1; do
        env RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_NT_COPY=$v timeout 120 python tools/probes/host_rate.py 1.5 2>/dev/null | sed "s/^RF_COPY_STREAMS=[0-9]*/RF_NT_COPY=$v rep $rep/" | tee -a $O/host_rate_nt_copy.txt
      done; done ;;
    bench_others)   # the dedicated lines of configs [2] and [3] (three regions of >= 1 s each)
      timeout 600 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline --host-seconds 0 --no-pipeline-trace > $O/bench_int8_0517_b32.json 2> $O/bench_0517.err
      timeout 600 python bench.py --height 896 --width 1280 --batch 1 --no-cpu-baseline --host-seconds 0 --no-pipeline-trace > $O/bench_1280x896_b1_fp16.json 2> $O/bench_1280.err
      for f in $O/bench_int8_0517_b32.json $O/bench_1280x896_b1_fp16.json; do python -c "import json;j=json.loads(open('$f').read().strip().splitlines()[-1]);r=j['roofline'];print('$(basename $f)', round(j['images_per_sec']), 'img/s', round(j['value']), 'faces/s', r['kernel_instance'], round(r['kernel_ms']*1e3,1), r['bound'], r['frac'])"; done ;;
    cpol)           # cache policy of the activation stores (build-time RF_STORE_CPOL: retinaface_amd/lib_cpol{2,16,17}) vs the default write-back, pipeline throughput + kernel sum, interleaved
      for rep in 1 2; do for which in ${CPOL_SET:-0 2 16 17}; do
        L=$R/retinaface_amd/lib/libretinaface_amd.so; [ $which != 0 ] && L=$R/retinaface_amd/lib_cpol$which/libretinaface_amd.so
        RETINAFACE_AMD_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-extra-configs --no-pmc --no-pipeline-trace --host-seconds 0 --regions 2 > $O/bench_cpol${which}_$rep.json 2> $O/bench_cpol${which}_$rep.err
        python -c "import json;j=json.loads(open('$O/bench_cpol${which}_$rep.json').read().strip().splitlines()[-1]);print('cpol $which rep $rep:', round(j['images_per_sec']), 'img/s  burst', round(j['burst']['ms'],3), 'ms  sync', round(j['sync_batch']['ms_per_call'],4), ' sum', round(j['roofline']['whole_path']['kernels_ms_per_launch_sequence']*1e3,1))" 2>&1 | tail -1
      done; done ;;
    coalesce)       # super-batch size: enqueued batch-8 tickets merged per launch (default 32 = 256 images), interleaved
      for rep in 1 2; do for c in ${COALESCE_SET:-32 24 20 28 40}; do
        timeout 200 python bench.py --coalesce $c --no-cpu-baseline --no-extra-configs --no-pmc --no-pipeline-trace --host-seconds 0 --regions 2 > $O/bench_coalesce${c}_$rep.json 2> $O/bench_coalesce${c}_$rep.err
        python -c "import json;j=json.loads(open('$O/bench_coalesce${c}_$rep.json').read().strip().splitlines()[-1]);print('coalesce $c rep $rep:', round(j['images_per_sec']), 'img/s  burst', round(j['burst']['ms'],3), 'ms for', j['burst']['images'], 'images')" 2>&1 | tail -1
      done; done ;;
    headstart)      # synchronous call latency with the first kernel launched eagerly ahead of the graph of the rest (RF_HEAD_START=1) vs one graph (0), probe build, interleaved
      for rep in 1 2 3; do for v in 0 1; do for b in 8 1; do
        env RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_HEAD_START=$v timeout 100 python tools/probes/sync_latency.py $b 1 2>/dev/null | grep "sync call" | sed "s/^/RF_HEAD_START=$v rep $rep: /" | tee -a $O/sync_latency_head_start.txt
      done; done; done ;;
    lds)            # LDS-array counters per kernel of the final tree (one --pmc pass, no trace flags)
      rm -rf /tmp/lds1; ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE -d /tmp/lds1 -o pmc -- python $R/tools/probes/pmc_probe.py 256 > $O/lds_probe.log 2>&1 )
      db=$(find /tmp/lds1 -name "*.db" | head -1); python tools/probes/lds_counters.py $db SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS > $O/lds_counters_fp16.txt 2>&1; cat $O/lds_counters_fp16.txt | cut -c1-140 ;;
    lanes)          # launches in flight: 2 / 3 / 4 lanes (default 3)
      for rep in 1 2 3; do for l in ${LANES_SET:-3 2 4}; do
        timeout 200 python bench.py --lanes $l --no-cpu-baseline --no-extra-configs --no-pmc --no-pipeline-trace --host-seconds 0 --regions 2 > $O/bench_lanes${l}_$rep.json 2> $O/bench_lanes${l}_$rep.err
        python -c "import json;j=json.loads(open('$O/bench_lanes${l}_$rep.json').read().strip().splitlines()[-1]);print('lanes $l rep $rep:', round(j['images_per_sec']), 'img/s')" 2>&1 | tail -1
      done; done ;;
    *) echo "unknown recipe $recipe" ;;
  esac
done
