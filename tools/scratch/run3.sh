set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for i in 1 2; do
DVD_GRU_INLAUNCH=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-prof > gpurun_out/r3/bench_old_$i.json 2> gpurun_out/r3/bench_old_$i.err
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-prof > gpurun_out/r3/bench_new_$i.json 2> gpurun_out/r3/bench_new_$i.err
done
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r3/bench_*.json
DVD_PROF_CSV=/tmp/shapes.csv timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/prof_shapes.py /tmp/shapes.csv 300 > gpurun_out/r3/shapes_all.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/gpu_tests.log
tail -5 gpurun_out/r3/gpu_tests.log
