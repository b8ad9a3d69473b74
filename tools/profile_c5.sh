#!/bin/bash
# kernel trace of the C5 leg alone (depth slabs engaged by the adaptive policy): gpurun_out/prof_c5_<tag>/stats/*kernel_stats.csv
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof_c5_$tag
rm -rf "$out"; mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 tools/config_leg.py C5 < /dev/null > "$out/stats.log" 2>&1
f=$(find "$out/stats" -name "*kernel_stats.csv" | head -1)
echo "== $f"; python3 tools/kstats.py "$f" 45
