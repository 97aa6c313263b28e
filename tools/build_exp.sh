#!/bin/bash
# builds tools/bin/exp_stream and tools/bin/exp_stream_probe (cross-compiles for gfx950 without a GPU)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
F="-DMI355_STREAM_SPIN_LIMIT=16384 --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -I cortex.llamacpp_amd/csrc"
SRC="tools/exp_stream.hip cortex.llamacpp_amd/csrc/mmvq.hip cortex.llamacpp_amd/csrc/mmvq_fast.hip cortex.llamacpp_amd/csrc/mmvq_stream.hip"
/opt/rocm/bin/hipcc $F $SRC -o tools/bin/exp_stream &
/opt/rocm/bin/hipcc $F -DMI355_STREAM_PROBE $SRC -o tools/bin/exp_stream_probe &
/opt/rocm/bin/hipcc $F -DMI355_ST_RING=65536 $SRC -o tools/bin/exp_stream_r64 &
for d in $EXP_DEPTHS; do
  /opt/rocm/bin/hipcc $F -DMI355_STREAM_DEPTH=$d $SRC -o tools/bin/exp_stream_d$d &
done
wait
