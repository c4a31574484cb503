R=$GRAFT_REPO_ROOT
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DLG_PHASE_CLOCKS -I$R/include -I$R/lidar-gs_amd/csrc $R/tools/micro/scatter_phases.cpp $R/lidar-gs_amd/csrc/binning.hip -o /tmp/scp && /tmp/scp
