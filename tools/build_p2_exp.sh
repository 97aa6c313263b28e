#!/bin/bash
# timing-experiment builds of the library: mmq.hip with -DMI355_P2_EXP=n (1 no fold, 2 no DMA, 3 no MFMA) -> tools/bin/libp2exp<n>.so
set -e
cd "$(dirname "$0")/.."
P=cortex.llamacpp_amd
python3 $P/build.py > /dev/null
mkdir -p tools/bin
for n in ${P2_EXPS:-1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -x hip -DMI355_P2_EXP=$n ${P2_DEFS} -c $P/csrc/mmq.hip -o /tmp/mmq_exp$n.o
  objs=$(ls $P/build/*.o | grep -v csrc_mmq.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o tools/bin/libp2exp$n.so $objs /tmp/mmq_exp$n.o -ldl
done
ls -la tools/bin/libp2exp*.so
