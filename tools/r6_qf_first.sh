#!/bin/bash
# round 6: first run of the Q | K | V-inside-the-attention-launch form: bitwise test, same-box A/B, probe time line
mkdir -p gpurun_out/qf
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "qkv_inside" > gpurun_out/qf/test.log 2>&1; echo "test rc=$?" >> gpurun_out/qf/test.log
tail -5 gpurun_out/qf/test.log
for i in 1 2; do
  MI355_QKV_ATTN_FUSED=1 timeout 300 python tools/time_decode.py 192 512 2> gpurun_out/qf/on_$i.err | tail -1 | sed 's/^/qf on : /'
  MI355_QKV_ATTN_FUSED=0 timeout 300 python tools/time_decode.py 192 512 2> gpurun_out/qf/off_$i.err | tail -1 | sed 's/^/qf off: /'
done
MI355_AO_PROBE=1 MI355_QKV_ATTN_FUSED=1 timeout 300 python tools/time_decode.py 8 512 2> gpurun_out/qf/probe_on.txt | tail -1
MI355_AO_PROBE=1 MI355_QKV_ATTN_FUSED=0 timeout 300 python tools/time_decode.py 8 512 2> gpurun_out/qf/probe_off.txt | tail -1
grep -A20 "attn_out probe" gpurun_out/qf/probe_on.txt | tail -22
