#!/bin/bash
# same-box A/B of decode tok/s between ENVIRONMENT settings of one library: tools/ab_env.sh name=VAR=VALUE [name=VAR=VALUE ...]; alternates REPS times
REPS=${REPS:-3}
mkdir -p gpurun_out
OUT=gpurun_out/ab_env.txt
: > $OUT
python3 tools/time_decode.py 8 > /dev/null 2>&1      # writes the model file once
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    name=${spec%%=*}; kv=${spec#*=}
    r=$(env "$kv" python3 tools/time_decode.py 192 2>/dev/null | tail -1)
    echo "$name $r" | tee -a $OUT
  done
done
python3 - <<'PY' | tee -a gpurun_out/ab_env.txt
import statistics, collections
d = collections.defaultdict(list)
for l in open('gpurun_out/ab_env.txt'):
    p = l.split()
    if len(p) == 4:
        d[p[0]].append((float(p[1]), float(p[2]), float(p[3])))
for k, v in d.items():
    print(f"median {k}: decode {statistics.median(x[0] for x in v):.1f}  filled {statistics.median(x[1] for x in v):.1f}  prefill512 {statistics.median(x[2] for x in v):.0f}   ({len(v)} runs)")
PY
