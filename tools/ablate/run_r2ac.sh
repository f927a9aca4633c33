# power / clock under the K >= 64 configs (is cfg3 on the power cap too?)
rm -f gpurun_out/pw6.log
for cfg in cfg3 cfg5 cfg2; do
  echo "== $cfg" >> gpurun_out/pw6.log
  VND_PROBE_CFG=$cfg timeout -k 10 120 python tools/power_probe.py 2 3 >> gpurun_out/pw6.log 2>&1 || exit 1
done
python tools/summarize_power.py gpurun_out/pw6.log
