#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355_PROFILER_SAFE=1
O=$PWD/gpurun_out/enc; rm -rf "$O"; mkdir -p "$O"
FT=${1:-f16}
python3 tools/bench_encoder.py $FT 512 4 > /dev/null 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/stats" -o q -- python3 "$OLDPWD/tools/bench_encoder.py" $FT 512 4 > "$O/bench.txt" 2> "$O/stats.err" )
DB=$(find "$O/stats" -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/prof_summary.py "$DB" "$O/kernel_stats.txt" | head -16
find "$O" -name "*.db" -size +20M -delete
