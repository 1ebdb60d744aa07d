#!/bin/bash
# Kernel trace + HBM traffic counters of one command on the GPU box (measurement tool).
#   tools/profile_run.sh <label> <python script> [args...]
# Three runs of the command: rocprofv3 --kernel-trace --stats, then one --pmc pass each for FETCH_SIZE and WRITE_SIZE
# (counters in runs of their own, never combined with a trace).  Results under gpurun_out/prof_<label>/;
# profiles/summarize_pmc.py turns the two counter files into bytes per launch.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$1
shift
mkdir -p "$O"
S=$R/$1
shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$O/stats" -o s --output-format csv -- python3 "$S" "$@" > "$O/stats.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$O/fetch" -o f --output-format csv -- python3 "$S" "$@" > "$O/fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$O/write" -o w --output-format csv -- python3 "$S" "$@" > "$O/write.log" 2>&1
find "$O" -name "*agent_info*" -delete
F=$(find "$O/fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$O/write" -name "*counter_collection.csv" | head -1)
python3 "$R/profiles/summarize_pmc.py" "$F" "$W" "$O/pmc_traffic.json"
K=$(find "$O/stats" -name "*kernel_stats.csv" | head -1)
cp "$K" "$O/kernel_stats.csv"
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*counter_collection.csv" -size +8M -delete
head -12 "$O/kernel_stats.csv"
