#!/bin/bash
# Round-3 profile collection on the GPU box (results under gpurun_out/r3prof/, summaries copied to profiles/ by hand afterwards):
#   1. rocprofv3 --kernel-trace --stats over the default bench command (eager launches: rocprofv3 7.2 crashes while tracing hipGraph replays)
#   2. rocprofv3 --pmc FETCH_SIZE over 16 decode steps  -> HBM read bytes per token of the mat-vec launches
#   3. two --pmc passes over a 512-token prompt        -> matrix-pipe busy / VALU per MFMA of the prompt kernels
#   4. rocm-smi power / clock samples while the prompt contraction runs back to back (tools/bin/exp_p2_e0: gate|up-sized launches, seconds of them)
# Counter passes carry --kernel-trace only (no sys / hip / memory-copy tracing next to --pmc).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1
O=$PWD/gpurun_out/r3prof
rm -rf "$O"; mkdir -p "$O"
python3 -c "import sys; sys.path.insert(0, '.'); import bench; print(bench.kernel_sources_sha256('.'))" > "$O/kernel_sources_sha256.txt" 2>/dev/null
python3 tools/decode_loop.py 1 8 > /dev/null 2>&1            # writes the synthetic model once, outside the profiled runs
echo "== 1 kernel stats"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/stats" -o r3 -- python3 "$OLDPWD/bench.py" --steps 128 --warmup 16 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/stats.err" )
DB=$(find "$O/stats" -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/prof_summary.py "$DB" "$O/r3_rocprof_kernel_stats.txt" | head -30
echo "== 2 decode traffic"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/fetch" -o r3 -- python3 "$OLDPWD/tools/decode_loop.py" 16 > /dev/null 2> "$O/fetch.err" )
DB=$(find "$O/fetch" -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/pmc_traffic.py "$DB" "$O/r3_pmc_fetch_size_by_kernel.json" | head -20
echo "== 3 prefill pmc"
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d "$O/pf1" -o r3 -- python3 "$OLDPWD/tools/decode_loop.py" 1 512 > /dev/null 2> "$O/pf1.err" )
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --kernel-trace -d "$O/pf2" -o r3 -- python3 "$OLDPWD/tools/decode_loop.py" 1 512 > /dev/null 2> "$O/pf2.err" )
D1=$(find "$O/pf1" -name "*_results.db" | head -1); D2=$(find "$O/pf2" -name "*_results.db" | head -1)
[ -n "$D1" ] && [ -n "$D2" ] && python3 tools/pmc_prefill.py "$D1" "$D2" "$O/r3_pmc_prefill_mfma.json" "round 3" | head -40
echo "== 4 power / clock under the prompt contraction"
unset MI355_NO_GRAPHS MI355_PROFILER_SAFE
( for i in $(seq 1 40); do echo "t=$i"; rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" ; sleep 0.25; done ) > "$O/r3_rocm_smi_prefill_trace.txt" 2>&1 &
SMI=$!
tools/bin/exp_p2_e0 28672 4096 2048 12 3000 > "$O/exp_p2_loop.txt" 2>&1
wait $SMI
head -12 "$O/r3_rocm_smi_prefill_trace.txt"
find "$O" -name "*.db" -size +20M -delete        # keep the pull small
ls -la "$O"
