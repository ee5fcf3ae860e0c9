#!/bin/bash
# Runs on the GPU box: matrix-pipe occupancy / clock of the dominant kernels on the microbenchmark shapes (rocprofv3 PMC pass; counters
# in tools/pmc_mfma.txt, no trace domains beside --kernel-trace) -> gpurun_out/<tag>_pmc_{conv,wgrad}.json
#   bash tools/collect_pmc.sh <tag>
set -u
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c_$tag /tmp/pmc_w_$tag
GB=1 rocprofv3 -i $root/tools/pmc_mfma.txt --kernel-trace -d /tmp/pmc_c_$tag -o p --output-format csv -- python $root/tools/conv_microbench.py fwd 3 3,4,5,12,0,1 > /dev/null 2>&1
python $root/tools/pmc_mfma_report.py /tmp/pmc_c_$tag $root/gpurun_out/${tag}_pmc_conv.json "GB=1 rocprofv3 -i tools/pmc_mfma.txt --kernel-trace -d /tmp/pmc_c -o p --output-format csv -- python tools/conv_microbench.py fwd 3 3,4,5,12,0,1"
rocprofv3 -i $root/tools/pmc_mfma.txt --kernel-trace -d /tmp/pmc_w_$tag -o p --output-format csv -- python $root/tools/conv_microbench.py wgrad 9 21,22,23,24,27,30,33,35 > /dev/null 2>&1
python $root/tools/pmc_mfma_report.py /tmp/pmc_w_$tag $root/gpurun_out/${tag}_pmc_wgrad.json "rocprofv3 -i tools/pmc_mfma.txt --kernel-trace -d /tmp/pmc_w -o p --output-format csv -- python tools/conv_microbench.py wgrad 9 21,22,23,24,27,30,33,35"
