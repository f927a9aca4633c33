# cfg4's N = 8 shard through CU chunks: the store-phase / prologue / priority switches one at a time and together
run() { echo "== $*"; env VND_TUNING=1 "$@" python tools/shard_timeline.py 128 600 2>&1 | grep -v amdgpu.ids | head -3 | cut -c1-200; }
run VND_WIN_CHUNKS=1
run VND_WIN_FILL_BATCH=24
run VND_WIN_RB_BATCH=4
run VND_WIN_RB_BATCH=8
run VND_WIN_SKIP_FINAL=1
run VND_WIN_CHUNK_PRIO=1
run VND_WIN_STAGGER_TICKS=300
run VND_WIN_STAGGER_TICKS=300 VND_WIN_CHUNK_PRIO=1
run VND_WIN_STAGGER_TICKS=300 VND_WIN_CHUNK_PRIO=1 VND_WIN_SKIP_FINAL=1 VND_WIN_RB_BATCH=4
run VND_WIN_STAGGER_TICKS=300 VND_WIN_CHUNK_PRIO=1 VND_WIN_SKIP_FINAL=1 VND_WIN_RB_BATCH=4 VND_WIN_FILL_BATCH=24
run VND_WIN_CHUNKS=1
