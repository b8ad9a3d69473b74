#!/bin/bash
# EXPERIMENT (needs the MSGS_X_OCC_GRID hook compiled in): occ_pass_kernel's time at C5 (big path, nothing closes) and at C3
# (early exit) for persistent grids of 256 / 512 / 1024 workgroups
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$root"
for g in 256 512 768 1024; do
  export MSGS_X_OCC_GRID=$g
  out=gpurun_out/occgrid_$g; rm -rf "$out"; mkdir -p "$out"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c5" -- python3 tools/config_leg.py C5 < /dev/null > "$out/c5.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c3" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pyramid --no-two-view --no-configs --no-kernel-timing < /dev/null > "$out/c3.log" 2>&1
  echo "== grid $g"
  python3 tools/kstats.py "$(find $out/c5 -name '*kernel_stats.csv' | head -1)" 60 | grep -E "occ_pass|slab_recount"
  python3 -c "import json,sys; d=json.loads([l for l in open('$out/c5.log') if l.startswith('{')][-1]); print('C5 ms', d['C5']['ms_per_step'])"
  python3 tools/kstats.py "$(find $out/c3 -name '*kernel_stats.csv' | head -1)" 60 | grep -E "occ_pass"
  tail -1 "$out/c3.log" | cut -c1-120
done
