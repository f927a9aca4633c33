# mono -> stereo, fast mode, merged plain form: reads ahead, chunk plan, priority of the store phase
run() { echo "== $*"; env VND_TUNING=1 "$@" python tools/fanout_win_try.py 2>&1 | grep "mode 2 automatic" | cut -c1-140; }
run VND_SPEC_LA=4
run VND_SPEC_LA=6
run VND_SPEC_LA=3
run VND_WIN_CHUNKS=0
run VND_WIN_RB_BATCH=4
run VND_SPEC_NT=128
run VND_SPEC_LA=4
