#!/bin/bash
# round 3: (a) the host API's time pieces for one long stream, (b) small shards of cfg4, one and two HIP streams
out=gpurun_out/r3c.log; : > $out
for p in 0 1 2 3 4 6; do
  if [ $p = 0 ]; then VND_HOST_TIME_CHUNKS=0 timeout -k 5 100 python tools/host_pieces_try.py >> $out 2>&1; else VND_HOST_TIME_PIECES=$p timeout -k 5 100 python tools/host_pieces_try.py >> $out 2>&1; fi
done
timeout -k 5 400 python tools/shard_try.py 128 256 1024 >> $out 2>&1
grep -v amdgpu.ids $out
