#!/bin/bash
# Round-6 profile collection on the GPU box (results under gpurun_out/r6prof/, turned into profiles/r6_* by tools/assemble_profiles_r6.py):
#   1. rocprofv3 --kernel-trace --stats over the default bench command (eager launches: rocprofv3 7.2 crashes while tracing hipGraph replays)
#   2. the same over 64 decode steps only (tools/decode_loop.py 64) -> one row per ROLE of the weight-stream kernel (Q|K|V, gate|up, ffn_down, head): roofline.frac per role
#   3. the same over ONE 512-token prompt and one step -> the prompt kernels by themselves
#   4. rocprofv3 --pmc FETCH_SIZE over 16 decode steps  -> HBM read bytes per token of the mat-vec launches
#   5. two --pmc passes over a 512-token prompt        -> matrix-pipe busy / VALU per MFMA of the prompt kernels
# Counter passes carry --kernel-trace only (no sys / hip / memory-copy tracing next to --pmc).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1
O=$PWD/gpurun_out/r6prof
rm -rf "$O"; mkdir -p "$O"
python3 -c "import sys; sys.path.insert(0, '.'); import bench; print(bench.kernel_sources_sha256('.'))" > "$O/kernel_sources_sha256.txt" 2>/dev/null
python3 tools/decode_loop.py 1 8 > /dev/null 2>&1            # writes the synthetic model once, outside the profiled runs
echo "== 1 kernel stats, bench command"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/stats" -o r6 -- python3 "$OLDPWD/bench.py" --steps 128 --warmup 16 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/stats.err" )
DB=$(find "$O/stats" -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/prof_summary.py "$DB" "$O/r6_rocprof_kernel_stats.txt" | head -12
echo "== 2 kernel stats, 64 decode steps"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/dec" -o r6 -- python3 "$OLDPWD/tools/decode_loop.py" 64 > /dev/null 2> "$O/dec.err" )
DB=$(find "$O/dec" -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/prof_summary.py "$DB" "$O/r6_rocprof_decode_kernel_stats.txt" | head -8
echo "== 3 kernel stats, one 512-token prompt"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/pre" -o r6 -- python3 "$OLDPWD/tools/decode_loop.py" 1 512 > /dev/null 2> "$O/pre.err" )
DB=$(find "$O/pre" -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/prof_summary.py "$DB" "$O/r6_rocprof_prefill_kernel_stats.txt" | head -24
echo "== 4 decode traffic"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/fetch" -o r6 -- python3 "$OLDPWD/tools/decode_loop.py" 16 > /dev/null 2> "$O/fetch.err" )
DB=$(find "$O/fetch" -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/pmc_traffic.py "$DB" "$O/r6_pmc_fetch_size_by_kernel.json" | head -12
echo "== 5 prefill pmc"
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d "$O/pf1" -o r6 -- python3 "$OLDPWD/tools/decode_loop.py" 1 512 > /dev/null 2> "$O/pf1.err" )
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --kernel-trace -d "$O/pf2" -o r6 -- python3 "$OLDPWD/tools/decode_loop.py" 1 512 > /dev/null 2> "$O/pf2.err" )
D1=$(find "$O/pf1" -name "*_results.db" | head -1); D2=$(find "$O/pf2" -name "*_results.db" | head -1)
[ -n "$D1" ] && [ -n "$D2" ] && python3 tools/pmc_prefill.py "$D1" "$D2" "$O/r6_pmc_prefill_mfma.json" "round 6" | head -30
unset MI355_NO_GRAPHS MI355_PROFILER_SAFE
echo "== 6 bench line (graphs), after the profiled runs"
python3 bench.py > "$O/r6_bench.json" 2> "$O/bench.err"; tail -c 1500 "$O/r6_bench.json"
find "$O" -name "*.db" -delete        # keep the pull small
ls -la "$O"
