#!/bin/bash
mkdir -p gpurun_out/qf
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "qkv_inside" 2>&1 | tail -3
MI355_AO_PROBE=1 MI355_QKV_ATTN_FUSED=1 timeout 300 python tools/r6_qf_probe.py 512 2> gpurun_out/qf/probe_on.txt | tail -1
MI355_AO_PROBE=1 MI355_QKV_ATTN_FUSED=0 timeout 300 python tools/r6_qf_probe.py 512 2> gpurun_out/qf/probe_off.txt | tail -1
MI355_AO_PROBE=1 MI355_QKV_ATTN_FUSED=1 timeout 300 python tools/r6_qf_probe.py 3968 2> gpurun_out/qf/probe_on_4k.txt | tail -1
grep -A20 "attn_out probe" gpurun_out/qf/probe_on.txt | tail -22
grep -A20 "attn_out probe" gpurun_out/qf/probe_off.txt | tail -22
