# far-first accumulation order (VND_WIN_FAR_FIRST) and pacing on the cfg4 shards: same box, interleaved
for rep in 1 2; do
for e in "VND_WIN_FAR_FIRST=1" "VND_WIN_FAR_FIRST=0" "VND_WIN_FAR_FIRST=1 VND_WIN_PACE=0" "VND_WIN_FAR_FIRST=0 VND_WIN_PACE=0"; do
  echo "== $e"
  for s in 256 1024; do env VND_TUNING=1 $e python tools/shard_timeline.py $s 200 2>&1 | grep "hipGraph" | sed "s/^/   $s streams: /"; done
done
done
