cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in dec_stereo_1 dec_mono_1 dec_stereo_128 dec_mono_128 dec_mono_512 conv_class_128 fn_batched_128 fn_mono_128 chain_1 c8_dec_16; do
  out=gpurun_out/audit_$s; rm -rf $out
  timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python3 tools/api_audit.py $s > gpurun_out/audit_$s.log 2>&1
  echo "== $s: $(grep 'ms per call' gpurun_out/audit_$s.log)"
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:7]:
    print(f"   {r['Name'][:90]:90s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
done
