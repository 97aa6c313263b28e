#!/bin/bash
# same-box A/B of the decode bench: the previous commit's tree (tools/bin/prevtree, built from `git archive`) against this tree, alternating
mkdir -p gpurun_out
for rep in 1 2 3; do
  for t in prev cur; do
    if [ $t = prev ]; then d=tools/bin/prevtree; else d=.; fi
    ( cd $d && python bench.py --steps 192 --warmup 16 --no-cpu-baseline --no-long-context 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$t rep $rep', d['value'], d.get('decode_tok_s_device_greedy'), d['prefill_tok_s'], d['roofline']['avg_launch_us'])
" )
  done
done 2>&1 | tee gpurun_out/ab_prev.txt
