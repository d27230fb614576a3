#!/bin/bash
# Host-side sanitizer run (CPU only, no GPU needed): builds retinaface_amd/lib/libretinaface_amd_asan.so (`make asan`: every host source under
# AddressSanitizer + UndefinedBehaviorSanitizer, kernels unsanitized) and runs the CPU tests that go through the C ABI against it -- model readers,
# graph compiler, weight packer (plan-cache probe), rf_convert_model / rf_attach_calibration / rf_plan_* entry points, corrupt-file rejection, the
# no-GPU failure path.  The tests that spawn gcc-sanitized binaries of their own, the bench dry runs and the knob subprocess test are left out (two
# sanitizer runtimes cannot share a process tree under LD_PRELOAD).        usage: bash tools/run_host_asan.sh [extra pytest args]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
PATH=/opt/rocm/bin:$PATH make -j8 -C "$R/retinaface_amd/csrc" asan > /dev/null
RT=$(dirname "$(find /opt/rocm*/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)")
cd "$R"
LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so LD_LIBRARY_PATH=$RT:$LD_LIBRARY_PATH ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
RETINAFACE_AMD_LIB=$R/retinaface_amd/lib/libretinaface_amd_asan.so \
python -m pytest tests/test_host.py tests/test_int8_oracle.py tests/test_oracle.py -q -m "not gpu" \
    -k "not sanitizers and not bench and not knob_table and not pack_index and not copier" "$@"
