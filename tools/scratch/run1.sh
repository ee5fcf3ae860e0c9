set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "convgru or generator" > gpurun_out/r1/t_modules.log 2>&1; echo "rc=$?" >> gpurun_out/r1/t_modules.log
tail -5 gpurun_out/r1/t_modules.log
DVD_GRU_TICKETS=0 timeout 300 python tools/gru_microbench.py 3 > gpurun_out/r1/gru_old.txt 2>&1
DVD_GRU_TICKETS=1 timeout 300 python tools/gru_microbench.py 3 > gpurun_out/r1/gru_new.txt 2>&1
paste -d'\n' gpurun_out/r1/gru_old.txt gpurun_out/r1/gru_new.txt
for i in 1 2; do
DVD_GRU_INLAUNCH=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-prof > gpurun_out/r1/bench_old_$i.json 2> gpurun_out/r1/bench_old_$i.err
DVD_GRU_INLAUNCH=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-prof > gpurun_out/r1/bench_new_$i.json 2> gpurun_out/r1/bench_new_$i.err
done
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r1/bench_*.json
