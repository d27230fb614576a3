cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline --host-seconds 0 > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err
python - <<'P'
import json
j = json.load(open("gpurun_out/c_bench.json"))
print("img/s %.0f faces/s %.0f steps %d ms/step %.4f burst %s" % (j["images_per_sec"], j["value"], j["steps"], j["ms_per_step"], j["burst"]))
r = j["roofline"]; print({k: r[k] for k in ("kernel_instance", "frac", "traffic", "traffic_source", "kernel_ms", "all_kernels_ms", "end_to_end_frac", "mfma_frac_all_kernels")})
P
