#!/bin/bash
# ISA / register report of one instantiation per kernel family of conv_gb.hip (seconds instead of minutes).
# usage: tools/isa_probe.sh [extra -D flags]; output under /tmp/isa_probe
set -e
mkdir -p /tmp/isa_probe && cd /tmp/isa_probe
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DDVD_GB_PROBE "$@" \
  -c /root/repo/dvd_gan_amd/csrc/conv_gb.hip -o probe.o -save-temps -Rpass-analysis=kernel-resource-usage 2> res.txt || { cat res.txt | grep -E "error" -A3; exit 1; }
grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill" res.txt | sed 's/ \[-Rpass.*//; s/.*remark: [^ ]* *//' | paste - - - - - - | cut -c1-260
