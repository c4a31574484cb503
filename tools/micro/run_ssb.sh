R=$GRAFT_REPO_ROOT
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I$R/include -I$R/lidar-gs_amd/csrc $R/tools/micro/small_sort_bench.cpp $R/lidar-gs_amd/csrc/binning.hip -o /tmp/ssb && /tmp/ssb
