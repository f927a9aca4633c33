# round 2, call A: LDS alignment microbenchmark, then packed add vs packed FMA under the power cap
mkdir -p gpurun_out; rm -f gpurun_out/pw2.log
timeout -k 10 120 tools/micro/lds_unaligned > gpurun_out/micro_lds.log 2>&1; echo "micro exit $?" >> gpurun_out/micro_lds.log
cat gpurun_out/micro_lds.log
for v in base ablate5 ablate2 base ablate5 ablate3; do
  unset VND_AMD_LIBRARY
  if [ $v != base ]; then export VND_AMD_LIBRARY=$PWD/tools/ablate/libvnd_$v.so; fi
  echo "== $v" >> gpurun_out/pw2.log
  timeout -k 10 120 python tools/power_probe.py 2 3 >> gpurun_out/pw2.log 2>&1 || exit 1
done
python tools/summarize_power.py gpurun_out/pw2.log
