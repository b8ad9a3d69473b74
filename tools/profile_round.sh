#!/bin/bash
# Round profile of bench.py on the GPU box: kernel trace + stats, and the PMC passes (each in its own run, no trace
# domains with --pmc).  Usage (through gpurun): bash tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/summary/*.csv
# Copy the summaries you want judged into profiles/ (rNN_kernel_stats.csv, rNN_pmc_*.csv, rNN_idle.txt) and summary/traffic.json,
# summary/sq.json to profiles/traffic_rNN.json, profiles/sq_rNN.json: bench.py reads those two while their csrc hash matches.
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out/summary"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pyramid --no-two-view --no-configs < /dev/null > "$out/stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pyramid --no-two-view --no-configs --no-kernel-timing < /dev/null > "$out/pmc_$c.log" 2>&1
done
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_sq" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pyramid --no-two-view --no-configs --no-kernel-timing < /dev/null > "$out/pmc_sq.log" 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/pmc_sq2" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pyramid --no-two-view --no-configs --no-kernel-timing < /dev/null > "$out/pmc_sq2.log" 2>&1
python3 tools/summarize_rocprof.py "$out" "$out/summary"
tail -1 "$out/stats.log" | cut -c1-400
ls -la "$out/summary"
