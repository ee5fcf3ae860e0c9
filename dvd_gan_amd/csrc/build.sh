#!/bin/bash
# Builds libdvdgan_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [-j N]
# Every compile also leaves build/<file>.res = hipcc's kernel-resource-usage remarks (registers, spills, scratch per kernel):
# tests/test_abi_cpu.py holds the hot kernels to "no scratch" with it (a dynamically indexed accumulator array once put every
# convolution kernel's accumulators into scratch memory: 3x on the step, invisible to every parity test).
set -e
cd "$(dirname "$0")"
JOBS=${JOBS:-6}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -Rpass-analysis=kernel-resource-usage"
mkdir -p build
pids=()
for f in *.hip; do
  o=build/${f%.hip}.o
  r=build/${f%.hip}.res
  if [ ! -f "$o" ] || [ ! -f "$r" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || { [[ "$f" == conv_* ]] && { [ conv_common.h -nt "$o" ] || [ prof.h -nt "$o" ]; }; } || [ ../../include/dvdgan_hip.h -nt "$o" ]; then
    ( hipcc $FLAGS -c "$f" -o "$o" 2> "$r.tmp" || { cat "$r.tmp" >&2; rm -f "$o" "$r.tmp"; exit 1; }
      grep -v "remark:\|^ *[0-9]* |\|^ *| *^" "$r.tmp" >&2 || true      # warnings stay visible
      grep "remark:" "$r.tmp" > "$r" || true; rm -f "$r.tmp" ) &
    pids+=($!)
    if [ ${#pids[@]} -ge $JOBS ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libdvdgan_hip.so build/*.o
echo "built $(pwd)/libdvdgan_hip.so"
