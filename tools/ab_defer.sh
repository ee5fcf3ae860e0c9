# A/B of the held-back weight gradients (DVD_SIDE_DEFER), the frame extent that releases them (DVD_DEFER_HW), a budget of held-back
# work (DVD_DEFER_TF, TFLOP; 0 = everything) and the CU mask
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-prof 2>&1 | grep -o '"ms_per_step": [0-9.]*'; }
run DVD_SIDE_DEFER=0
run DVD_SIDE_DEFER=1 DVD_DEFER_HW=8 DVD_DEFER_TF=5
run DVD_SIDE_DEFER=1 DVD_DEFER_HW=8 DVD_DEFER_TF=10
run DVD_SIDE_DEFER=1 DVD_DEFER_HW=8 DVD_DEFER_TF=20
run DVD_SIDE_DEFER=1 DVD_DEFER_HW=16 DVD_DEFER_TF=10
run DVD_SIDE_DEFER=1 DVD_DEFER_HW=16 DVD_DEFER_TF=25
run DVD_SIDE_DEFER=1 DVD_DEFER_HW=4 DVD_DEFER_TF=5
run DVD_SIDE_DEFER=0
