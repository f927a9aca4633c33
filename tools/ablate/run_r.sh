# sustained (power-capped) rate per tile size: the short interleaved bursts of tools/sweep.py run before the cap bites
mkdir -p gpurun_out; rm -f gpurun_out/rsweep.log
for v in 4 8 6 3 4 8; do
  echo "== pairs_per_lane $v" >> gpurun_out/rsweep.log
  timeout -k 10 120 python tools/sustain.py 2 $v >> gpurun_out/rsweep.log 2>&1
done
grep -E "==|launches +(200|275)" gpurun_out/rsweep.log
