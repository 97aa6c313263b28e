#!/bin/bash
# Round 6, VERDICT item 1 "done" check: the isolated 8-rank reduce-scatter + all-gather case ten times, each in a fresh pytest process, first thing on a fresh box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CASE='tests/test_gpu_tp.py::test_prompt_sized_exchange_as_reduce_scatter_all_gather[tiny-70b-2l-q4_k_m-q8_0-8-0]'
: > gpurun_out/r6_tp_isolated10.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  s=$(date +%s)
  MI355_TP_FRESH_PROCESS=1 timeout 600 python -m pytest "$CASE" -x -q -p no:cacheprovider > gpurun_out/r6_tp_iso_$i.log 2>&1
  rc=$?
  e=$(date +%s)
  echo "run $i: rc=$rc $(tail -1 gpurun_out/r6_tp_iso_$i.log) wall $((e - s)) s" | tee -a gpurun_out/r6_tp_isolated10.txt
  if [ $rc -ne 0 ]; then grep -n "gave up\|Error" gpurun_out/r6_tp_iso_$i.log | head -5 | cut -c1-300 | tee -a gpurun_out/r6_tp_isolated10.txt; fi
done
