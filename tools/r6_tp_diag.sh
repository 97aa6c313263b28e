#!/bin/bash
# Round 6: the isolated 8-rank reduce-scatter case in fresh pytest processes until it fails once (at most N runs, default 10), short exchange bound, trace +
# the flags that never arrived.  VERDICT r5 item 1's "done": 10 of 10.
cd "$(dirname "$0")/.."
N=${1:-10}
mkdir -p gpurun_out; rm -f gpurun_out/tp_trace_*
CASE='tests/test_gpu_tp.py::test_prompt_sized_exchange_as_reduce_scatter_all_gather[tiny-70b-2l-q4_k_m-q8_0-8-0]'
: > gpurun_out/r6_tp_isolated10.txt
for i in $(seq 1 $N); do
  s=$(date +%s)
  MI355_TP_TRACE=1 MI355_TP_YIELD_TIMEOUT_S=8 MI355_TP_FRESH_PROCESS=1 timeout 600 python -m pytest "$CASE" -x -q -p no:cacheprovider > gpurun_out/r6_tp_diag_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/r6_tp_diag_$i.log) wall $(( $(date +%s) - s )) s" | tee -a gpurun_out/r6_tp_isolated10.txt
  if [ $rc -ne 0 ]; then break; fi
  rm -f gpurun_out/tp_trace_*
done
grep -h "^\[tp\]\|==== rank" gpurun_out/tp_trace_tp-out-rsag* 2>/dev/null | cut -c1-500 | head -40
