mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4prof
rm -rf $O; mkdir -p $O
python3 tools/decode_loop.py 1 8 > /dev/null 2>&1
( cd /tmp && MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1 rocprofv3 --kernel-trace --stats -d $O/stats -o r4 -- python3 $R/tools/decode_loop.py 64 > /dev/null 2> $O/stats.err )
DB=$(find $O/stats -name "*_results.db" | head -1)
python3 tools/prof_summary.py $DB $O/r4_kernel_stats_decode_loop.txt > /dev/null
python3 tools/prof_timeline.py $DB --last 8000 $O/r4_timeline_decode_loop.txt > /dev/null
find $O -name "*.db" -delete
head -30 $O/r4_kernel_stats_decode_loop.txt
cat $O/r4_timeline_decode_loop.txt | head -40
