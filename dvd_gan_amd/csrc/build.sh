#!/bin/bash
# Builds libdvdgan_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [-j N]
set -e
cd "$(dirname "$0")"
JOBS=${JOBS:-6}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
mkdir -p build
pids=()
for f in *.hip; do
  o=build/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || { [[ "$f" == conv_* ]] && { [ conv_common.h -nt "$o" ] || [ prof.h -nt "$o" ]; }; } || [ ../../include/dvdgan_hip.h -nt "$o" ]; then
    ( hipcc $FLAGS -c "$f" -o "$o" ) &
    pids+=($!)
    if [ ${#pids[@]} -ge $JOBS ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libdvdgan_hip.so build/*.o
echo "built $(pwd)/libdvdgan_hip.so"
