set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "convgru or generator" > gpurun_out/r2/t_modules.log 2>&1; echo "rc=$?" >> gpurun_out/r2/t_modules.log
tail -3 gpurun_out/r2/t_modules.log
run() { name=$1; shift; env "$@" timeout 300 python tools/gru_microbench.py 3 0,1,2,3,4,5,6,7,8 > gpurun_out/r2/gru_$name.txt 2>&1; }
run off DVD_GRU_INLAUNCH=0
run d2 DVD_GRU_INLAUNCH=2
run d4 DVD_GRU_INLAUNCH=4
run cap2 DVD_NS_CAP=2 DVD_GRU_INLAUNCH=2
run cap4 DVD_NS_CAP=4 DVD_GRU_INLAUNCH=4
run cap4off DVD_NS_CAP=4 DVD_GRU_INLAUNCH=0
for f in off d2 d4 cap2 cap4 cap4off; do echo "== $f"; grep -v amdgpu.ids gpurun_out/r2/gru_$f.txt | sed 's/ns(ur,o,dur)=//' | cut -c1-160; done
