cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-prof 2>&1 | grep -o '"ms_per_step": [0-9.]*'; }
run A=1
run DVD_SIDE_CUS=128 DVD_SIDE_CUPAT=0
run DVD_SIDE_CUS=128 DVD_SIDE_CUPAT=1
run DVD_SIDE_CUS=128 DVD_SIDE_CUPAT=2
run DVD_SIDE_CUS=192 DVD_SIDE_CUPAT=0
run DVD_SIDE_CUS=192 DVD_SIDE_CUPAT=1
run DVD_SIDE_CUS=64 DVD_SIDE_CUPAT=1
run A=1
