# cfg4's N = 4 shard (256 streams: 6 tiles per CU): uniform 3 + 3 against forced CU chunks 4 + 2 / 5 + 1, and the stagger
for env in "VND_WIN_CHUNKS=0" "VND_WIN_CHUNK_LEN0=3" "VND_WIN_CHUNK_LEN0=4" "VND_WIN_CHUNK_LEN0=5" "VND_WIN_CHUNK_LEN0=4 VND_WIN_STAGGER_TICKS=300" "VND_WIN_CHUNK_LEN0=3 VND_WIN_STAGGER_TICKS=300"; do
  echo "== 256 streams $env"
  env VND_TUNING=1 $env python tools/shard_timeline.py 256 300 2>&1 | grep -v amdgpu.ids | head -3
done
for env in "VND_WIN_CHUNKS=0" "VND_WIN_CHUNKS=1" "VND_WIN_STAGGER_TICKS=200" "VND_WIN_STAGGER_TICKS=250" "VND_WIN_STAGGER_TICKS=300" "VND_WIN_STAGGER_TICKS=350" "VND_WIN_STAGGER_TICKS=400"; do
  echo "== 128 streams $env"
  env VND_TUNING=1 $env python tools/shard_timeline.py 128 600 2>&1 | grep -v amdgpu.ids | head -3
done
