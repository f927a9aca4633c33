# interleaved A/B of a diagnostic build against the shipped library (sustained rate, one process each)
mkdir -p gpurun_out; rm -f gpurun_out/aux.log
for v in base s2 base s2 base s2; do
  if [ $v = base ]; then unset VND_AMD_LIBRARY; else export VND_AMD_LIBRARY=$PWD/tools/ablate/libvnd_$v.so; fi
  echo "== $v" >> gpurun_out/aux.log
  timeout -k 10 120 python tools/sustain.py 2 >> gpurun_out/aux.log 2>&1
done
grep -E "==|launches +(200|275)|python loop" gpurun_out/aux.log
