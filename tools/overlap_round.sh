#!/bin/bash
# kernel traces of the serial and the two-stream loops (through gpurun): bash tools/overlap_round.sh <tag> [modes...]
tag=${1:-x}; shift
modes=${@:-serial piped piped_shared fwd_serial fwd_piped_shared}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/ovl_$tag
rm -rf "$out"; mkdir -p "$out"
for m in $modes; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out/$m" -- python3 tools/two_view_loop.py $m 4 8 < /dev/null > "$out/$m.log" 2>&1
  f=$(find "$out/$m" -name '*kernel_trace.csv' | head -1)
  echo "== $m: $(grep ms/view $out/$m.log)" | tee -a "$out/summary.txt"
  python3 tools/overlap_trace.py "$f" 2>&1 | tee -a "$out/summary.txt"
  rm -rf "$out/$m"
done
