# round 2, call B: board power / clock of the specialised kernel vs the generic one, then PMC passes
mkdir -p gpurun_out; rm -f gpurun_out/pw3.log
for v in spec generic spec512 spec128; do
  unset VND_SPEC VND_SPEC_NT VND_SPEC_RR VND_SPEC_DD
  [ $v = generic ] && export VND_SPEC=0
  [ $v = spec512 ] && export VND_SPEC_NT=512 VND_SPEC_RR=2 VND_SPEC_DD=3
  [ $v = spec128 ] && export VND_SPEC_NT=128 VND_SPEC_RR=4
  echo "== $v" >> gpurun_out/pw3.log
  timeout -k 10 120 python tools/power_probe.py 2 3 >> gpurun_out/pw3.log 2>&1 || exit 1
done
python tools/summarize_power.py gpurun_out/pw3.log
unset VND_SPEC VND_SPEC_NT VND_SPEC_RR VND_SPEC_DD
bash tools/profile.sh r02_spec > gpurun_out/profile_r02_spec.log 2>&1
tail -5 gpurun_out/profile_r02_spec.log
