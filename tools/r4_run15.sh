# down-projection with ready-made codes (separate quantiser launch) against the quantising prologue: per-kernel time under rocprofv3, eager launches
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1
O=$PWD/gpurun_out/r4_fd; rm -rf "$O"; mkdir -p "$O"
python3 tools/decode_loop.py 1 8 > /dev/null 2>&1
for v in 1 0; do
  export MI355_FUSE_DOWN=$v
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/s$v" -o r4 -- python3 "$OLDPWD/tools/decode_loop.py" 64 > /dev/null 2> "$O/s$v.err" )
  DB=$(find "$O/s$v" -name "*_results.db" | head -1)
  echo "== MI355_FUSE_DOWN=$v"; python3 tools/prof_summary.py "$DB" "$O/stats_fuse_down_$v.txt" | head -14
done
find "$O" -name "*.db" -delete
