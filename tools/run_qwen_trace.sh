#!/bin/bash
# kernel trace of the Qwen2-7B-geometry bench (eager launches), summary under gpurun_out/qwen/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1
O=$PWD/gpurun_out/qwen; rm -rf "$O"; mkdir -p "$O"
CFG=${1:-qwen2-7b}
python3 bench.py --config $CFG --ftype q4_k_m --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2>&1     # writes the model once
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/stats" -o q -- python3 "$OLDPWD/bench.py" --config $CFG --ftype q4_k_m --steps 64 --warmup 8 --no-cpu-baseline > "$O/bench.json" 2> "$O/stats.err" )
DB=$(find "$O/stats" -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/prof_summary.py "$DB" "$O/kernel_stats.txt" | head -40
find "$O" -name "*.db" -size +20M -delete
