#!/bin/bash
# kernel trace of the headline step only: gpurun_out/prof_c3_<tag>/stats/*kernel_stats.csv
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof_c3_$tag
rm -rf "$out"; mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pyramid --no-two-view --no-configs --no-kernel-timing < /dev/null > "$out/stats.log" 2>&1
tail -1 "$out/stats.log" | cut -c1-300
