mkdir -p gpurun_out
MI355_NO_GRAPHS=1 MI355_AO_PROBE=1 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_ao_probe.txt
MI355_NO_GRAPHS=1 MI355_ATTN_OUT_FUSED=0 MI355_ATTN_PROBE=1 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_attn_probe_old.txt
MI355_NO_GRAPHS=1 MI355_AO_PROBE=1 python bench.py --steps 4 --warmup 2 --prompt 3960 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_ao_probe_long.txt
tail -12 gpurun_out/r4_ao_probe.txt gpurun_out/r4_attn_probe_old.txt gpurun_out/r4_ao_probe_long.txt
