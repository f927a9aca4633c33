mkdir -p gpurun_out; rm -f gpurun_out/pw.log
for v in ablate2 ablate2b128 base fullb128; do
  m=2; unset VND_AMD_LIBRARY
  if [ $v != base ]; then export VND_AMD_LIBRARY=$PWD/tools/ablate/libvnd_$v.so; fi
  echo "== $v" >> gpurun_out/pw.log
  timeout -k 10 120 python tools/power_probe.py $m 3 >> gpurun_out/pw.log 2>&1 || exit 1
done
python tools/summarize_power.py gpurun_out/pw.log
