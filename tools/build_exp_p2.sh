#!/bin/bash
# builds tools/bin/exp_p2 and tools/bin/exp_p2_probe (cross-compiles for gfx950 without a GPU); P2_DEFS adds -D flags
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -I cortex.llamacpp_amd/csrc $P2_DEFS"
SRC="tools/exp_p2.hip cortex.llamacpp_amd/csrc/mmq.hip"
/opt/rocm/bin/hipcc $F $SRC -o tools/bin/exp_p2${P2_SUFFIX} &
/opt/rocm/bin/hipcc $F -DMI355_P2_PROBE $SRC -o tools/bin/exp_p2_probe${P2_SUFFIX} &
wait
