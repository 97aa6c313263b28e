#!/bin/bash
# round 5: variants of the weight-stream harness (tools/exp_stream.hip + csrc/mmvq_stream.hip); mmvq.hip / mmvq_fast.hip compiled once.
# usage: tools/build_exp_r5.sh name[:extra-flags] ...   e.g.  base  probe:-DMI355_STREAM_PROBE  noearly:-DMI355_STREAM_EARLY_ACT=0
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin /tmp/exp_r5
F="-DMI355_STREAM_SPIN_LIMIT=16384 --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -I cortex.llamacpp_amd/csrc"
PRE="-mllvm -amdgpu-kernarg-preload-count=14"
[ -f /tmp/exp_r5/mmvq.o ] || /opt/rocm/bin/hipcc $F -c cortex.llamacpp_amd/csrc/mmvq.hip -o /tmp/exp_r5/mmvq.o &
[ -f /tmp/exp_r5/mmvq_fast.o ] || /opt/rocm/bin/hipcc $F -c cortex.llamacpp_amd/csrc/mmvq_fast.hip -o /tmp/exp_r5/mmvq_fast.o &
wait
build() {   # name, extra flags
  /opt/rocm/bin/hipcc $F $2 -c tools/exp_stream.hip -o /tmp/exp_r5/exp_$1.o
  /opt/rocm/bin/hipcc $F $PRE $2 -c cortex.llamacpp_amd/csrc/mmvq_stream.hip -o /tmp/exp_r5/stream_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/exp_r5/exp_$1.o /tmp/exp_r5/stream_$1.o /tmp/exp_r5/mmvq.o /tmp/exp_r5/mmvq_fast.o -o tools/bin/exp5_$1
}
for v in "$@"; do
  name=${v%%:*}; flags=""
  [ "$name" != "$v" ] && flags=${v#*:}
  build "$name" "$flags" &
done
wait
ls -la tools/bin/exp5_*
