#!/bin/bash
# round 4: variants of the weight-stream harness (consumer wave count, row pairs for long rows); mmvq.hip / mmvq_fast.hip compiled once
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin /tmp/exp_r4
F="-DMI355_STREAM_SPIN_LIMIT=16384 --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -I cortex.llamacpp_amd/csrc"
[ -f /tmp/exp_r4/mmvq.o ] || /opt/rocm/bin/hipcc $F -c cortex.llamacpp_amd/csrc/mmvq.hip -o /tmp/exp_r4/mmvq.o &
[ -f /tmp/exp_r4/mmvq_fast.o ] || /opt/rocm/bin/hipcc $F -c cortex.llamacpp_amd/csrc/mmvq_fast.hip -o /tmp/exp_r4/mmvq_fast.o &
wait
build() {   # name, extra flags
  /opt/rocm/bin/hipcc $F $2 -c tools/exp_stream.hip -o /tmp/exp_r4/exp_$1.o
  /opt/rocm/bin/hipcc $F $2 -c cortex.llamacpp_amd/csrc/mmvq_stream.hip -o /tmp/exp_r4/stream_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/exp_r4/exp_$1.o /tmp/exp_r4/stream_$1.o /tmp/exp_r4/mmvq.o /tmp/exp_r4/mmvq_fast.o -o tools/bin/exp_stream_$1
}
for v in "$@"; do
  case $v in
    base) build base "" & ;;
    nc14) build nc14 "-DMI355_ST_NC=14" & ;;
    nc12) build nc12 "-DMI355_ST_NC=12" & ;;
    pair) build pair "-DMI355_ST_PAIR_MAX=24576" & ;;
    nc14pair) build nc14pair "-DMI355_ST_NC=14 -DMI355_ST_PAIR_MAX=24576" & ;;
    probe) build probe "-DMI355_STREAM_PROBE" & ;;
    nomins) build nomins "-DMI355_EXP_NO_MINS" & ;;
    early0) build early0 "-DMI355_STREAM_EARLY=0" & ;;
    early4) build early4 "-DMI355_STREAM_EARLY=4" & ;;
    nc14probe) build nc14probe "-DMI355_ST_NC=14 -DMI355_STREAM_PROBE" & ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
wait
