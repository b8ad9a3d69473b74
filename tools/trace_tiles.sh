#!/bin/bash
# Tool build of the library with per-tile timestamps in the one-wave-per-tile backward (-DMSGS_TRACE_TILES) and one traced
# C3 backward: writes gpurun_out/<tag>/tile_trace.npy.  Usage (through gpurun): bash tools/trace_tiles.sh <tag> [env ...]
tag=${1:-trace}; shift
root=${GRAFT_REPO_ROOT:-$PWD}; cd "$root"
mkdir -p gpurun_out/$tag /tmp/trace_build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -DMSGS_TRACE_TILES"
for f in api preprocess sort binning blend voxel_pool epilogue loss knn; do
  /opt/rocm/bin/hipcc $F -c ms-gs_amd/csrc/$f.hip -o /tmp/trace_build/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/trace_build/libmsgs_hip_trace.so /tmp/trace_build/*.o
env "$@" MSGS_HIP_LIB=/tmp/trace_build/libmsgs_hip_trace.so python3 tools/trace_tiles.py gpurun_out/$tag
