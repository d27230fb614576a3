#!/bin/bash
# Collect the rocprofv3 evidence of one engine configuration on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh TAG IMAGES_PER_LAUNCH PRECISION MODEL H W BATCH [sq]
# -> gpurun_out/TAG_pmc_hbm_traffic.json (FETCH_SIZE x2 + WRITE_SIZE per kernel, separate passes as MI355X_MICROARCH.md prescribes)
#    gpurun_out/TAG_pmc_sq.json          (with "sq": two SQ counter passes + GRBM_GUI_ACTIVE)
# Counter passes never carry a trace flag (gpurun refuses --pmc together with runtime / sys traces).
set -u
TAG=$1; N=$2; PREC=$3; MODEL=$4; H=$5; W=$6; B=$7; SQ=${8:-}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # name counters...
    local name=$1; shift
    rm -rf $R/gpurun_out/${TAG}_$name
    rocprofv3 --pmc "$@" -d $R/gpurun_out/${TAG}_$name -o pmc -- python $R/tools/probes/pmc_probe.py $N $PREC $MODEL $H $W $B > $R/gpurun_out/${TAG}_$name.log 2>&1
    find $R/gpurun_out/${TAG}_$name -name "*.db" | head -1
}
F=$(run fetch FETCH_SIZE)
Wd=$(run write WRITE_SIZE)
python $R/tools/pmc_summary.py "$F" "$Wd" $R/gpurun_out/${TAG}_pmc_hbm_traffic.json $N ${H}x${W}_${PREC} | tail -3
if [ -n "$SQ" ]; then
    S1=$(run sq1 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)
    S2=$(run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT)
    (cd $R/tools && python pmc_sq_summary.py "$S1" "$S2" $R/gpurun_out/${TAG}_pmc_sq.json $N)
fi
# the raw databases are large: keep the summaries only
rm -rf $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write $R/gpurun_out/${TAG}_sq1 $R/gpurun_out/${TAG}_sq2
