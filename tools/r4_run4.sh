mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "attn_step" 2>&1 | tail -5) > gpurun_out/r4_t4_ops.log
(timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "attn_out_one_launch or argmax_follows" 2>&1 | tail -30) > gpurun_out/r4_t4_model.log
python bench.py --steps 128 --warmup 16 --no-cpu-baseline > gpurun_out/r4_bench_fused3.json 2> gpurun_out/r4_bench_fused3.err
MI355_NO_GRAPHS=1 MI355_AO_PROBE=1 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_ao_probe3.txt
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_a -o a -- $R/tools/bin/exp_stream_base 2 > $R/gpurun_out/pmc_a.out 2> $R/gpurun_out/pmc_a.err )
( cd /tmp && rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_b -o b -- $R/tools/bin/exp_stream_base 2 > $R/gpurun_out/pmc_b.out 2> $R/gpurun_out/pmc_b.err )
for x in a b; do f=$(find gpurun_out/pmc_$x -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_kernel.py $f stream > gpurun_out/r4_pmc_stream_$x.txt; done
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
