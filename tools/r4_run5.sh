mkdir -p gpurun_out
rm -f gpurun_out/r4_exp_planes.txt
for v in base nomins base nomins; do
  echo "=== $v" >> gpurun_out/r4_exp_planes.txt
  timeout 300 tools/bin/exp_stream_$v 20 2>&1 | grep -E "^qkv   q4k 4096\|1024\|1024 x4096 rmsnorm  |^o     q4k 4096x4096 planes \+resid      |^gateup|^down|layer chain" | grep -v differ >> gpurun_out/r4_exp_planes.txt
done
cat gpurun_out/r4_exp_planes.txt
