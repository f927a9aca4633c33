set -e
mkdir -p gpurun_out
rm -f gpurun_out/ab4.log
for v in base ablate4 s2 base ablate4; do
  if [ $v = base ]; then unset VND_AMD_LIBRARY; else export VND_AMD_LIBRARY=$PWD/tools/ablate/libvnd_$v.so; fi
  echo "== $v" >> gpurun_out/ab4.log
  timeout -k 10 120 python tools/sustain.py 2 >> gpurun_out/ab4.log 2>&1
done
grep -E "==|launches +(200|275)" gpurun_out/ab4.log
