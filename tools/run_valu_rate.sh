# VALU issue-rate calibration on the GPU box (tools/micro/valu_rate.hip): wall-clock rates + one SQ counter pass.
#   bash tools/run_valu_rate.sh r03      -> gpurun_out/r03_valu_rate.json, gpurun_out/r03_valu_rate_pmc.json
P=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $R/tools/micro/valu_rate.hip -o /tmp/valu_rate || exit 1
timeout 600 /tmp/valu_rate > $R/gpurun_out/${P}_valu_rate.json || exit 1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_ANY \
    -d /tmp/vr_pmc -o vr -- /tmp/valu_rate pmc > /tmp/vr_pmc.log 2>&1
DB=$(ls /tmp/vr_pmc/*/*.db /tmp/vr_pmc/*.db 2>/dev/null | head -1)
python $R/tools/valu_rate_pmc.py "$DB" > $R/gpurun_out/${P}_valu_rate_pmc.json
tail -c 600 $R/gpurun_out/${P}_valu_rate_pmc.json
