#!/bin/bash
# round 6: ONE parametrised script for every GPU call of the round (as tools/gpu/archive/r5.sh did for round 5).
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
    new_tests)      # the tests this round added / touched
      ( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "split_upload or rehearsal or concurrent or composition_invariant or two_ranks" ) > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
      grep -E "passed|failed|Error|assert|^rc|real" $O/pytest_new.log | tail -20 ;;
    multi_rehearsal)  # the N > 1 line on the one-GPU box: 2 ranks sharing the GPU (gloo), library leg over 8 ordinals with forced scatter
      timeout 600 python bench.py --gpus 2 --oversubscribe --no-cpu-baseline --host-seconds 0 --no-pmc --no-pipeline-trace --profile-iters 5 --ring-mb 80 --min-seconds 0.5 \
          > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err; echo "2 ranks rc $?"
      timeout 300 python bench.py --library-devices 8 > $O/bench_library_leg_8_engines_one_gpu.json 2> $O/bench_library_leg.err; echo "library leg rc $?"
      python - <<PY
import json
j=json.loads(open("$O/bench_2ranks_one_gpu.json").read().strip().splitlines()[-1])
print("2 ranks:", round(j["images_per_sec"]), "img/s; efficiency", j.get("scaling_efficiency",{}).get("value"), "per rank", j["result_gather"]["per_rank"]["images_per_sec"])
print(" split_ab", {k:round(v["ms_per_step"],3) for k,v in j["batch_split_ab"]["legs"].items()}, "same", j["batch_split_ab"]["same_detections_every_leg"])
l=j.get("library_multi_device",{})
print(" library leg", l.get("ms_per_call"), l.get("leg_seconds"), l.get("error"))
l=json.loads(open("$O/bench_library_leg_8_engines_one_gpu.json").read().strip().splitlines()[-1])["library_multi_device"]
print("library 8:", round(l["ms_per_call"],3), "ms; identical", l["detections_identical_to_single_engine"], "leg s", round(l["leg_seconds"],1))
print(" split_ab", {k:(v if not isinstance(v,dict) else {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()}) for k,v in l["split_ab"].items() if k!="what"})
PY
      ;;
    sync_pieces)    # piece-count sweep of the pipelined staging (RF_SYNC_PIECES), batch 8 and 32 at 448 x 448, one 1280 x 896 frame
      for pc in 1 2 4 8 16; do RF_SYNC_PIECES=$pc timeout 120 python tools/probes/sync_host_pieces.py; done > $O/sync_host_pieces.txt 2>&1; cat $O/sync_host_pieces.txt ;;
    sync_host)      # one synchronous host-frame call, pipelined staging vs RF_SYNC_SPLIT=0, every config
      timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-pipeline-trace --profile-iters 5 > $O/bench_sync_host.json 2> $O/bench_sync_host.err; echo "rc $?"
      python - <<PY
import json
j=json.loads(open("$O/bench_sync_host.json").read().strip().splitlines()[-1])
print("value", round(j["images_per_sec"]), "img/s")
for c in [dict(id="main", sync_batch_host=j.get("sync_batch_host"), sync_batch=j["sync_batch"])] + j.get("configs", []):
    h=c.get("sync_batch_host")
    if h: print(" cfg", c["id"], "device", round(c["sync_batch"]["ms_per_call"],4), "pageable", round(h["pageable"]["ms_per_call"],4), h["pageable"]["byte_identical_to_device_frames"], "registered", round(h["registered"]["ms_per_call"],4), h["registered"]["byte_identical_to_device_frames"], "unsplit", h.get("unsplit"))
PY
      ;;
    stem_ab)        # stem2 raw-row staging (V2 = 15) vs round 5's product (V2 = 7), probe build, interleaved; then the tests that look at the stem
      for rep in 1 2; do for v in 7 15; do
        RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 256 --tag r6_${TAG}_stemv$v > $O/kbench_stem_v${v}_$rep.txt 2>&1
        grep -E "total|stem2" $O/kbench_stem_v${v}_$rep.txt | cut -c1-90 | sed "s/^/V2=$v rep $rep: /"
      done; done
      for v in 7 15; do RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 64 --hw 896 1280 --tag r6_${TAG}_stemv${v}_big 2>&1 | grep -E "total|stem2" | cut -c1-90 | sed "s/^/V2=$v 1280x896: /"; done
      ( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "fused_op or head_blobs or unaligned or network_presets or fp16_contract or oversize or pad32 or smoke" ) > $O/pytest_stem.log 2>&1; echo "rc $?" >> $O/pytest_stem.log
      grep -E "passed|failed|Error|assert|^rc|real|contract" $O/pytest_stem.log | cut -c1-400 | tail -20 ;;
    insts)          # dynamic instruction mix per kernel (SQ_INSTS_* / SQ_WAVES): the product, and stem2 of round 5 (V2 = 7, probe build) beside it
      ( cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_[A-Z_0-9]*" | sort -u | tr "\n" " " ) > $O/sq_insts_counters_available.txt; cat $O/sq_insts_counters_available.txt; echo
      for v in product 7 int8; do
        rm -rf /tmp/im$v
        E=""; W="256"; M=SQ_INSTS_VALU_MFMA_MOPS_F16
        if [ $v = 7 ]; then E="RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_STEM2_V2=$v"; fi
        if [ $v = int8 ]; then W="256 int8 mnet25 448 448 32"; M=SQ_INSTS_VALU_MFMA_MOPS_I8; fi
        ( cd /tmp && env $E timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS $M SQ_WAVES -d /tmp/im$v -o pmc -- python $R/tools/probes/pmc_probe.py $W > $O/insts_$v.log 2>&1 )
        db=$(find /tmp/im$v -name "*.db" | head -1)
        if [ -z "$db" ]; then ( cd /tmp && env $E timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/im$v -o pmc -- python $R/tools/probes/pmc_probe.py $W >> $O/insts_$v.log 2>&1 ); db=$(find /tmp/im$v -name "*.db" | head -1); fi
        echo "--- $v"; python tools/pmc_insts.py $db $O/instruction_mix_$v.json | head -20
      done ;;
    stem8_ab)       # int8 stem: raw-row staging (RF_STEM_RAW=1, default) vs the general path, probe build, interleaved
      for rep in 1 2; do for v in 0 1; do
        RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_STEM_RAW=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r6_${TAG}_stem8raw$v 2>&1 | grep -E "total|stem " | cut -c1-90 | sed "s/^/RAW=$v rep $rep: /"
      done; done ;;
    probes)         # the measured-and-rejected kernel variants of the probe build, held to the default path's parity bar
      make -C retinaface_amd/csrc probe > /dev/null 2>&1
      ( time RF_PROBE_TESTS=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k probe_knob ) > $O/pytest_probe_knobs.log 2>&1; tail -4 $O/pytest_probe_knobs.log ;;
    trace1)         # single-lane rocprofv3 kernel trace of the timed loop (fp16 metric point, then int8 configs[4] shape)
      rm -rf /tmp/kt1; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt -- python $R/bench.py --timed-only --lanes 1 --no-pmc > $O/trace1_bench.json 2> $O/trace1.err )
      db=$(find /tmp/kt1 -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_fp16.txt > /dev/null; head -24 $O/kernel_trace_lanes1_fp16.txt | cut -c1-60,104-190
      rm -rf /tmp/kt2; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt -- python $R/bench.py --timed-only --lanes 1 --no-pmc --precision int8 --batch 32 > $O/trace1_int8_bench.json 2> $O/trace1_int8.err )
      db=$(find /tmp/kt2 -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/kernel_trace_lanes1_int8.txt > /dev/null; head -26 $O/kernel_trace_lanes1_int8.txt | cut -c1-60,104-190 ;;
    bench_int8)     # the dedicated lines of configs[4] (per-GPU shape) and configs[2]
      timeout 400 python bench.py --precision int8 --model mnet25 --batch 32 --no-cpu-baseline --no-extra-configs > $O/bench_int8_mnet25_b32.json 2> $O/bench_int8_mnet25.err
      timeout 400 python bench.py --precision int8 --model mnet-deconv-0517 --batch 32 --no-cpu-baseline --no-extra-configs > $O/bench_int8_0517_b32.json 2> $O/bench_int8_0517.err
      for f in bench_int8_mnet25_b32 bench_int8_0517_b32; do python -c "import json;j=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f', round(j['images_per_sec']), 'img/s', j['roofline']['kernel_instance'], round(j['roofline']['kernel_ms']*1e3,1), 'us')"; done ;;
    inst_classes)   # VALU instruction classes per kernel (conversion / integer / fp arithmetic / MFMA), product library, fp16 then int8
      for v in fp16 int8; do
        W="256"; if [ $v = int8 ]; then W="256 int8 mnet25 448 448 32"; fi
        for pass in a b; do
          rm -rf /tmp/ic$v$pass
          if [ $pass = a ]; then C="SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_MFMA SQ_WAVES"; else C="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_WAVES"; fi
          ( cd /tmp && timeout 300 rocprofv3 --pmc $C -d /tmp/ic$v$pass -o pmc -- python $R/tools/probes/pmc_probe.py $W > $O/inst_classes_${v}_$pass.log 2>&1 )
          db=$(find /tmp/ic$v$pass -name "*.db" | head -1)
          echo "--- $v pass $pass"; python tools/pmc_insts.py $db $O/instruction_classes_${v}_$pass.json | head -8
        done
      done ;;
    stem_bits)      # which of stem2's index-from-memory bits pays: 15 base | 47 expanded conv3 fragments | 31 conv0 table | 95 table + explicit ds_read2_b32 | 127 all
      for rep in 1 2; do for v in 15 47 31 95 127; do
        RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 256 --tag r6_${TAG}_stemv$v 2>&1 | grep -E "stem2" | cut -c1-60 | sed "s/^/V2=$v rep $rep: /"
      done; done
      RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so timeout 400 python tools/probes/knob_equal.py --precision 1 --n 16 RF_STEM2_V2=47 RF_STEM2_V2=95 RF_STEM2_V2=127 2>&1 | tail -3
      for rep in 1 2; do for v in 1 2; do
        RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_STEM_RAW=$v timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag r6_${TAG}_stem8raw$v 2>&1 | grep -E "stem " | cut -c1-60 | sed "s/^/int8 RAW=$v rep $rep: /"
      done; done ;;
    gridfrac)       # persistent grids sized for a FRACTION of the resident slots, so that another lane's kernel can be co-resident (RF_GRID_FRAC, probe build), x lanes
      for rep in 1 2; do for cfg in "1.0 3" "0.75 3" "0.5 3" "0.5 4" "0.5 6" "0.67 4" "0.34 6" "1.0 3"; do set -- $cfg
        env RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_GRID_FRAC=$1 timeout 200 python bench.py --lanes $2 --no-cpu-baseline --no-extra-configs --no-pmc --no-pipeline-trace --host-seconds 0 --regions 2 --profile-iters 3 > $O/bench_frac$1_lanes$2_$rep.json 2> $O/bench_frac$1_lanes$2_$rep.err
        python -c "import json;j=json.loads(open('$O/bench_frac$1_lanes$2_$rep.json').read().strip().splitlines()[-1]);print('frac $1 lanes $2 rep $rep:', round(j['images_per_sec']), 'img/s  burst', round(j['burst']['ms'],3), 'ms  sync', round(j['sync_batch']['ms_per_call'],4))" 2>&1 | tail -1
      done; done ;;
    wide128)        # fp16 plain 128-channel blocks: K_b 4x8 (product) | K_b 4x16 | K_b(8) 4x8 | K_b(8) 4x16 (probe build)
      for rep in 1 2; do for cfg in "-1 1" "2 1" "-1 2" "2 2"; do set -- $cfg
        RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_TILE128=$1 RF_WIDE128=$2 timeout 200 python tools/kbench.py --n 256 --tag r6_${TAG}_t$1_w$2 2>&1 | grep -E "total|dwpw<128,128,s1>" | head -2 | cut -c1-80 | sed "s/^/TILE128=$1 WIDE128=$2 rep $rep: /"
      done; done
      RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so timeout 300 python tools/probes/knob_equal.py --precision 1 --n 16 "RF_TILE128=2" 2>&1 | tail -1 ;;
    upmerge)        # (round 6, rejected and removed from engine.cpp: see profiles/r06_sync_host_call_ab.txt (d); the knob no longer exists)
                    # pipelined host frames (rf_enqueue_batch, pageable): one DMA per enqueue (0) vs merged uploads of >= N MB (RF_UPLOAD_MERGE_MB)
      for rep in 1 2; do for mb in 0 8 16 32 64 0; do
        RF_UPLOAD_MERGE_MB=$mb timeout 200 python bench.py --no-cpu-baseline --no-extra-configs --no-pmc --no-pipeline-trace --host-seconds 3 --profile-iters 3 --regions 1 --min-seconds 0.3 > $O/bench_upmerge${mb}_$rep.json 2> $O/bench_upmerge${mb}_$rep.err
        python -c "import json;j=json.loads(open('$O/bench_upmerge${mb}_$rep.json').read().strip().splitlines()[-1]);h=j['host_frames'];print('merge $mb MB rep $rep: pageable', round(h['pageable']['images_per_sec']), 'img/s', round(h['pageable']['pcie_GBs'],1), 'GB/s | registered', round(h['registered']['images_per_sec']), '| pinned copy', round(h['pinned_copy_GBs_measured'],1), 'GB/s | device-frame value', round(j['images_per_sec']))" 2>&1 | tail -1
      done; done ;;
    driver_cmd)     # the driver's exact round-end commands, timed
      ( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_exact.json 2> $O/bench_driver_exact.err ) 2> $O/bench_driver_exact_time.txt; tail -3 $O/bench_driver_exact_time.txt
      python -c "import json;j=json.loads(open('$O/bench_driver_exact.json').read().strip().splitlines()[-1]);print('driver exact:', round(j['images_per_sec']), 'img/s', round(j['value']), j['unit'], 'steps', j['steps'], 'requested', j['steps_requested'], 'roofline.frac', j['roofline']['frac'], 'cpu_baseline', j['cpu_baseline']['value'], j['cpu_baseline']['unit'], 'instruction_mix valu', j['roofline']['instruction_mix']['per_wave']['valu'])" ;;
    stress)         # soak test of the scheduler: random interleavings of every entry point on one handle, every result compared byte for byte
      for cfg in "fp16 1 8" "int8 2 32" "fp16 3 1"; do set -- $cfg
        timeout 400 python tools/probes/stress.py --seconds ${STRESS_S:-60} --precision $1 --seed $2 --max-batch $3 2>&1 | tail -2
      done | tee $O/stress.txt
      for cfg in "int8 4 8 4" "fp16 5 8 3"; do set -- $cfg      # ONE handle over several engines (multi.cpp), forced scatter
        timeout 400 python tools/probes/stress.py --seconds ${STRESS_S:-60} --precision $1 --seed $2 --max-batch $3 --devices $4 2>&1 | tail -2
      done | tee -a $O/stress.txt
      timeout 400 python tools/probes/stress.py --threads 3 --seconds ${STRESS_S:-60} --precision fp16 --seed 7 --max-batch 8 2>&1 | grep -E "stress|Error|error" | tee -a $O/stress.txt ;;      # three threads, three handles, one GPU
    ranks8)         # the whole `--gpus 8` flow with REAL engines, eight ranks sharing the one GPU (gloo; RCCL refuses duplicates): launcher, legs, accounting
      ( time timeout 900 python bench.py --gpus 8 --oversubscribe --no-cpu-baseline --host-seconds 0 --no-pmc --no-pipeline-trace --profile-iters 3 --ring-mb 40 --min-seconds 0.3 --regions 1 \
          > $O/bench_8ranks_one_gpu.json 2> $O/bench_8ranks_one_gpu.err ) 2> $O/bench_8ranks_time.txt; echo "8 ranks rc $?"; tail -3 $O/bench_8ranks_time.txt
      python - <<PY
import json
j=json.loads(open("$O/bench_8ranks_one_gpu.json").read().strip().splitlines()[-1])
g=j["result_gather"]
print("8 ranks:", round(j["images_per_sec"]), "img/s; efficiency", round(j["scaling_efficiency"]["value"],3), "records", g["records_gathered"], "==", g["expected"], "ranks", g["ranks_in_communicator"])
print(" per rank", [round(x) for x in g["per_rank"]["images_per_sec"]])
print(" split_ab", {k:round(v["ms_per_step"],3) for k,v in j["batch_split_ab"]["legs"].items()}, "same", j["batch_split_ab"]["same_detections_every_leg"])
print(" strong", round(j["configs4_strong"]["images_per_sec"]), "img/s records", j["configs4_strong"]["result_gather"]["records_gathered"], "==", j["configs4_strong"]["result_gather"]["expected"])
l=j["library_multi_device"]; print(" library leg", l.get("ms_per_call"), "identical", l.get("detections_identical_to_single_engine"), l.get("error"))
PY
      ;;
    stem_tab)       # stem2 with index tables (V2 = 31, the product) vs raw staging alone (V2 = 15), probe build, interleaved; bit-identity of the two
      for rep in 1 2; do for v in 15 31; do
        RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 256 --tag r6_${TAG}_stemv$v > $O/kbench_stem_v${v}_$rep.txt 2>&1
        grep -E "total|stem2" $O/kbench_stem_v${v}_$rep.txt | cut -c1-90 | sed "s/^/V2=$v rep $rep: /"
      done; done
      for v in 15 31; do RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so RF_STEM2_V2=$v timeout 200 python tools/kbench.py --n 64 --hw 896 1280 --tag r6_${TAG}_stemv${v}_big 2>&1 | grep -E "total|stem2" | cut -c1-90 | sed "s/^/V2=$v 1280x896: /"; done
      RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_probe.so timeout 300 python tools/probes/knob_equal.py --precision 1 --n 16 RF_STEM2_V2=15 RF_STEM2_V2=31 > $O/knob_equal_stem_tab.txt 2>&1; tail -4 $O/knob_equal_stem_tab.txt ;;
    *) echo "unknown recipe $recipe" ;;
  esac
done
