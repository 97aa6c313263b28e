#!/bin/bash
# tools/build_lib_variant.sh <name> "<extra -D flags>" <source.hip> [<source2.hip> ...]: the library with the given sources recompiled under extra flags
# (all other objects from the tree's last build) -> tools/bin/libs/<name>.so   (for tools/ab_libs.sh)
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2; shift 2
mkdir -p tools/bin/libs /tmp/libvar_$name
B=cortex.llamacpp_amd/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -x hip"
objs=""
# (the objects the tree's library was linked from: build.py writes the list - the build directory may hold objects of the other experiments setting)
for o in $(cat $B/linked_objects.txt); do
  keep=1
  for s in "$@"; do
    if [ "$(basename $o)" = "$(echo $s | sed 's|/|_|g').o" ]; then keep=0; fi
  done
  [ $keep = 1 ] && objs="$objs $o"
done
for s in "$@"; do
  obj=/tmp/libvar_$name/$(echo $s | sed 's|/|_|g').o
  /opt/rocm/bin/hipcc $F $extra -c cortex.llamacpp_amd/$s -o $obj
  objs="$objs $obj"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o tools/bin/libs/$name.so $objs -ldl
echo tools/bin/libs/$name.so
