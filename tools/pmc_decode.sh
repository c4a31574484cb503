# SQ counters of the anchor decode's kernels (forward + backward at 666 667 anchors x 6), one pass per counter group (a pass whose
# counter name the box does not know fails alone):  bash tools/pmc_decode.sh <prefix>
P=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU SQ_BUSY_CYCLES"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_INSTS_FLAT"
n=0
DBS=""
for G in "$G1" "$G2" "$G3"; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G -d $R/gpurun_out/${P}_dec_sq$n -o t -- python $R/tools/time_decode.py 666667 6 4 hip > $R/gpurun_out/${P}_dec_sq$n.log 2>&1
  if [ -f $R/gpurun_out/${P}_dec_sq$n/t_results.db ]; then DBS="$DBS gpurun_out/${P}_dec_sq$n/t_results.db"; else tail -3 $R/gpurun_out/${P}_dec_sq$n.log; fi
done
cd $R
python tools/pmc_sq.py decode "rocprofv3 --kernel-trace --pmc <groups of tools/pmc_decode.sh> -- python tools/time_decode.py 666667 6 4 hip" $DBS > gpurun_out/${P}_pmc_sq_decode.json
rm -rf gpurun_out/${P}_dec_sq1 gpurun_out/${P}_dec_sq2 gpurun_out/${P}_dec_sq3
python - <<PY
import json
d=json.load(open("gpurun_out/${P}_pmc_sq_decode.json"))
for k,v in d["kernels"].items():
    if "ng_" in k: print(k, json.dumps(v))
PY
