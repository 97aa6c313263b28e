#!/bin/bash
# same-box A/B of the decode bench: the round-2 tree (tools/bin/r2tree, built from its commit) against this tree, alternating
mkdir -p gpurun_out
for rep in 1 2; do
  for t in r2 r3; do
    if [ $t = r2 ]; then d=tools/bin/r2tree; else d=.; fi
    ( cd $d && python bench.py --steps 192 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$t rep $rep', d['value'], d.get('decode_tok_s_device_greedy'), d['prefill_tok_s'], d['roofline']['avg_launch_us'], d.get('long_context', {}).get('decode_tok_s'))
" )
  done
done 2>&1 | tee gpurun_out/ab.txt
