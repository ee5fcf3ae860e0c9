#!/bin/bash
# Builds an experimental copy of the library with extra -D flags:
#   tools/build_variant.sh <name> "<-DFLAG ...>" ["file1.hip file2.hip ..."]  -> ab_libs/<name>.so
# The listed sources (default: conv_igemm.hip conv_gb.hip gru.hip) are recompiled with the flags; the other objects come from
# csrc/build (run build.sh first).
set -e
name=$1; flags=$2; files=${3:-"conv_igemm.hip conv_gb.hip gru.hip"}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/dvd_gan_amd/csrc
mkdir -p "$root/ab_libs/obj_$name"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
for f in $files; do
  ( cd "$src" && hipcc $F $flags -c "$f" -o "$root/ab_libs/obj_$name/${f%.hip}.o" ) &
done
wait
objs=""
for o in "$src"/build/*.o; do
  b=$(basename "$o")
  if [ -f "$root/ab_libs/obj_$name/$b" ]; then objs="$objs $root/ab_libs/obj_$name/$b"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/ab_libs/$name.so" $objs
echo "built ab_libs/$name.so"
