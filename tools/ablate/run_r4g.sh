# cfg5's octet kernel: the tile's first barrier dropped (a wave is the only reader of its channel's planes), read-backs in batches
run() { echo "== $*"; python tools/cfg5_try.py "$@" 2>&1 | grep "^cfg"; }
run VND_WIN_QC_SKIP_X=0 VND_WIN_RB_BATCH=1
run VND_WIN_QC_SKIP_X=1 VND_WIN_RB_BATCH=1
run VND_WIN_QC_SKIP_X=1 VND_WIN_RB_BATCH=4
run VND_WIN_QC_SKIP_X=1 VND_WIN_RB_BATCH=8
run VND_WIN_QC_SKIP_X=1 VND_WIN_RB_BATCH=8 VND_WIN_PRIO=0
run VND_WIN_QC_SKIP_X=0 VND_WIN_RB_BATCH=1
