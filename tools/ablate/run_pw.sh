mkdir -p gpurun_out; rm -f gpurun_out/pw.log
for v in copy ablate1 base exact; do
  m=2; unset VND_AMD_LIBRARY
  if [ $v = copy ]; then m=copy; elif [ $v = exact ]; then m=0; elif [ $v != base ]; then export VND_AMD_LIBRARY=$PWD/tools/ablate/libvnd_$v.so; fi
  echo "== $v" >> gpurun_out/pw.log
  timeout -k 10 120 python tools/power_probe.py $m 3 >> gpurun_out/pw.log 2>&1 || exit 1
done
python tools/summarize_power.py gpurun_out/pw.log
