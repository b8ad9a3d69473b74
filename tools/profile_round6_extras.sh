#!/bin/bash
# Round-6 extras next to tools/profile_round.sh: the C5 leg, the filters-off render and the training iteration under the kernel
# trace, and the host floor of a step.  Usage (through gpurun): bash tools/profile_round6_extras.sh -> gpurun_out/r6x/*
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/r6x; rm -rf "$out"; mkdir -p "$out"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c5" -- python3 tools/config_leg.py C5 < /dev/null > "$out/c5.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/off" -- python3 tools/time_filters_off.py 0 < /dev/null > "$out/off.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train" -- python3 tools/time_train_iteration.py C3 < /dev/null > "$out/train.log" 2>&1
for d in c5 off train; do cp "$(find $out/$d -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_$d.csv"; python3 tools/kstats.py "$out/kernel_stats_$d.csv" 30 > "$out/kernel_stats_$d.txt"; done
timeout 300 python3 tools/cpu_profile_step.py > "$out/host_floor_reference_api.txt" 2>&1
timeout 300 python3 tools/cpu_profile_step.py fused > "$out/host_floor_fused.txt" 2>&1
timeout 300 python3 tools/time_train_iteration.py C3 > "$out/train_iteration_ab.txt" 2>&1
grep -h "ms/step" "$out"/host_floor_*.txt; tail -4 "$out/train_iteration_ab.txt"; grep -E "^k=0" "$out/off.log" | cut -c1-400
