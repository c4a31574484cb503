R=$GRAFT_REPO_ROOT
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I$R/include -I$R/lidar-gs_amd/csrc $R/tools/micro/lookback_scatter.hip $R/lidar-gs_amd/csrc/binning.hip -o /tmp/lbs && timeout 120 /tmp/lbs
