#!/bin/bash
# Round 6: the decode calls' roctx ranges (MI355_ROCTX=1, host/runtime.cc TraceRange) as rocprofv3 sees them: --marker-trace beside --kernel-trace (no counters)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355_ROCTX=1 MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1
O=$PWD/gpurun_out/r6roctx; rm -rf "$O"; mkdir -p "$O"
python3 tools/decode_loop.py 1 8 > /dev/null 2>&1
( cd /tmp && rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$O/t" -o r6 -- python3 "$OLDPWD/tools/decode_loop.py" 8 > /dev/null 2> "$O/err.txt" )
find "$O" -name "*marker*" | head; for f in $(find "$O" -name "*marker_api_stats*.csv" -o -name "*marker*stats*.csv" | head -2); do echo "== $f"; head -12 "$f"; done
f=$(find "$O" -name "*marker_api_trace*.csv" | head -1); [ -n "$f" ] && { echo "== $f"; head -14 "$f" | cut -c1-220; }
tail -3 "$O/err.txt"
