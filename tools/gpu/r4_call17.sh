#!/bin/bash
# round 4, GPU call 17: the library compiled with -mllvm --amdgpu-mfma-vgpr-form (MFMA results land in VGPRs: no v_accvgpr_read copies in the epilogues): identity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c17
mkdir -p $O
cd $R
VF=$R/retinaface_amd/lib_vf/libretinaface_amd.so
timeout 600 python tools/probes/knob_equal.py --precision 2 RETINAFACE_AMD_LIB=$VF > $O/equal_int8.txt 2>&1
timeout 600 python tools/probes/knob_equal.py --precision 1 RETINAFACE_AMD_LIB=$VF > $O/equal_fp16.txt 2>&1
for rep in 1 2 3; do
  timeout 200 python tools/kbench.py --n 256 --tag fp16_base_$rep > $O/kbench_fp16_base_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$VF timeout 200 python tools/kbench.py --n 256 --tag fp16_vf_$rep > $O/kbench_fp16_vf_$rep.txt 2>&1
  timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_base_$rep > $O/kbench_int8_base_$rep.txt 2>&1
  RETINAFACE_AMD_LIB=$VF timeout 200 python tools/kbench.py --n 256 --precision int8 --batch 32 --tag int8_vf_$rep > $O/kbench_int8_vf_$rep.txt 2>&1
done
cat $O/equal_int8.txt $O/equal_fp16.txt
for f in $O/kbench_*.txt; do echo "$(basename $f) $(grep -h '==' $f | awk '{print $8}') | $(grep -v '==' $f | grep us | awk '{printf "%s ", $2}')"; done
