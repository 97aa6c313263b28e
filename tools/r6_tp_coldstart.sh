#!/bin/bash
# Round 6, VERDICT item 1: the 8-rank reduce-scatter + all-gather case as the FIRST multi-rank test of a fresh box, in a fresh pytest process, with the
# exchange trace on (host/tp_comm.cc MI355_TP_TRACE=1); then the same case again (warm) a few times.  Everything lands in gpurun_out/r6_tp_cold_*.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CASE='tests/test_gpu_tp.py::test_prompt_sized_exchange_as_reduce_scatter_all_gather[tiny-70b-2l-q4_k_m-q8_0-8-0]'
for i in 1 2 3; do
  ( time MI355_TP_TRACE=1 MI355_TP_FRESH_PROCESS=1 timeout 600 python -m pytest "$CASE" -x -q 2>&1 | grep -v "hostname of the client socket\|amdgpu.ids\|connected to 7 peer" | tail -150 ) > gpurun_out/r6_tp_cold_$i.log 2>&1
  echo "run $i: $(grep -c 'passed' gpurun_out/r6_tp_cold_$i.log) passed-lines; $(tail -4 gpurun_out/r6_tp_cold_$i.log | tr '\n' ' ')"
done
ls -la gpurun_out/tp_trace_* 2>/dev/null | head
