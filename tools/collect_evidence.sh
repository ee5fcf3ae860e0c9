#!/bin/bash
# Runs on the GPU box (through gpurun): the measurements profiles/ is built from.
#   bash tools/collect_evidence.sh <tag>      -> gpurun_out/ev_<tag>/{bench.json,stats.csv,shapes.txt,traffic.json}
set -u
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ev_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout 500 python bench.py > $out/bench.json 2> $out/bench.err
tail -c 1500 $out/bench.json
# kernel durations with every kernel alone on the GPU (weight gradients on the launch stream): the per-kernel roofline table
DVD_SIDE_SERIAL=1 timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o st -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-prof > $out/prof.log 2>&1
python tools/rocpd_summary.py /tmp/prof_$tag/st_results.db $out/stats.csv
tail -1 $out/prof.log | cut -c1-160
# the same command as it runs by default (weight gradients on the concurrent side stream: durations overlap)
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/profc_$tag -o st -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-prof > $out/prof_concurrent.log 2>&1
python tools/rocpd_summary.py /tmp/profc_$tag/st_results.db $out/stats_concurrent.csv
tail -1 $out/prof_concurrent.log | cut -c1-160
DVD_PROF_CSV=/tmp/shapes_$tag.csv timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/prof_shapes.py /tmp/shapes_$tag.csv 60 > $out/shapes.txt
timeout 900 rocprofv3 -i tools/pmc_traffic.txt --kernel-trace -d /tmp/pmc_$tag -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-prof > $out/pmc.log 2>&1
python tools/traffic_summary.py /tmp/pmc_$tag $out/traffic.json "build $tag"
python tools/stats_categories.py $out/stats.csv 3 > $out/categories.txt; cat $out/categories.txt
# other shapes / modes on the same build (one JSON line each)
for args in "--batch 32" "--batch 16" "--size 128 --n-class 600" "--frames 12 --size 128 --state-carry" "--g-attn both"; do
  timeout 600 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-prof 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d['_note']='tools/collect_evidence.sh $tag: bench.py $args --steps 5 --warmup 2'; print(json.dumps(d))" >> $out/other_lines.jsonl
done
cut -c1-200 $out/other_lines.jsonl
python tools/host_ahead_probe.py 16 6 > $out/host_b16.txt 2>&1; tail -2 $out/host_b16.txt
python tools/gru_microbench.py stack > $out/gru_stack.txt 2>&1; tail -5 $out/gru_stack.txt
python tools/sample_bench.py > $out/sample_path.txt 2>&1; tail -3 $out/sample_path.txt
