#!/bin/bash
# Builds an experimental copy of the library with extra -D flags: tools/build_variant.sh <name> "<-DFLAG ...>"  -> ab_libs/<name>.so
# conv_gb.hip / conv_igemm.hip / gru.hip are recompiled with the flags; the other objects come from csrc/build (run build.sh first).
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/dvd_gan_amd/csrc
mkdir -p "$root/ab_libs/obj_$name"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
( cd "$src" && hipcc $F $flags -c conv_igemm.hip -o "$root/ab_libs/obj_$name/conv_igemm.o" ) &
( cd "$src" && hipcc $F $flags -c conv_gb.hip -o "$root/ab_libs/obj_$name/conv_gb.o" ) &
( cd "$src" && hipcc $F $flags -c gru.hip -o "$root/ab_libs/obj_$name/gru.o" ) &
wait
objs=""
for o in "$src"/build/*.o; do
  b=$(basename "$o")
  if [ "$b" = conv_igemm.o ] || [ "$b" = gru.o ] || [ "$b" = conv_gb.o ]; then objs="$objs $root/ab_libs/obj_$name/$b"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/ab_libs/$name.so" $objs
echo "built ab_libs/$name.so"
