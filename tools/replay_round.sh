# Per-rank replay of the sharded frame at world 8 (tools/time_shell.py: compute + host glue per rank, no RCCL time), both cuts, both gradient forms:
#   bash tools/replay_round.sh <prefix>      -> gpurun_out/<prefix>_time_<cut>_<cfg>_<grad_sync>.txt
P=${1:-x}; R=$GRAFT_REPO_ROOT; cd $R
for gs in reduce_scatter shard; do
  GRAD_SYNC=$gs timeout 600 python tools/time_shell.py 8 cfg3 60 > gpurun_out/${P}_time_shell_cfg3_$gs.txt 2>&1
  MODE=wedge GRAD_SYNC=$gs timeout 600 python tools/time_shell.py 8 cfg3 60 > gpurun_out/${P}_time_wedge_cfg3_$gs.txt 2>&1
  MODE=wedge GRAD_SYNC=$gs timeout 600 python tools/time_shell.py 8 cfg4 60 > gpurun_out/${P}_time_wedge_cfg4_$gs.txt 2>&1
done
grep -H "per-rank ms\|collectives" gpurun_out/${P}_time_*.txt | cut -c1-260
