#!/bin/bash
# round 3, GPU call 16 (probe): does v_cvt_pk_u8_f32 round to nearest even by itself?  The bit-exact int8 test on a build without v_rndne_f32
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c16
mkdir -p $O
cd $R
RETINAFACE_AMD_LIB=$R/retinaface_amd/lib_dev/libretinaface_amd.so timeout 600 python -m pytest tests -m gpu -q -x -k "int8_engine_is_bit_exact" > $O/pytest_no_rndne.log 2>&1
grep -v "compute time" $O/pytest_no_rndne.log | grep -E "passed|failed|AssertionError|assert " | head -6
