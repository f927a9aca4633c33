mkdir -p gpurun_out; rm -f gpurun_out/pw.log
for v in ablate1 ablate2 ablate3 base exact; do
  m=2
  if [ $v = base ]; then unset VND_AMD_LIBRARY; elif [ $v = exact ]; then unset VND_AMD_LIBRARY; m=0; else export VND_AMD_LIBRARY=$PWD/tools/ablate/libvnd_$v.so; fi
  echo "== $v" >> gpurun_out/pw.log
  timeout -k 10 120 python tools/power_probe.py $m 3 >> gpurun_out/pw.log 2>&1 || exit 1
done
tail -3 gpurun_out/pw.log | cut -c1-200
