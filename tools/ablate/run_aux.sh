set -e
mkdir -p gpurun_out
for v in base s2 l2 s2l2 s1 s17 s3 base; do
  if [ $v = base ]; then unset VND_AMD_LIBRARY; else export VND_AMD_LIBRARY=$PWD/tools/ablate/libvnd_$v.so; fi
  echo "== $v" >> gpurun_out/aux.log
  timeout -k 10 120 python tools/sustain.py 2 >> gpurun_out/aux.log 2>&1
done
grep -E "==|launches (200|275)|python loop" gpurun_out/aux.log
