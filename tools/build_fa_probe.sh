#!/bin/bash
# probe build of the library: attn_prefill.hip with -DMI355_FA_PROBE -> tools/bin/libfaprobe.so (load with MI355_LLAMA_LIB)
set -e
cd "$(dirname "$0")/.."
P=cortex.llamacpp_amd
python3 $P/build.py > /dev/null
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -x hip -DMI355_FA_PROBE -c $P/csrc/attn_prefill.hip -o /tmp/fa_probe.o
objs=$(ls $P/build/*.o | grep -v csrc_attn_prefill.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o tools/bin/libfaprobe.so $objs /tmp/fa_probe.o -ldl
