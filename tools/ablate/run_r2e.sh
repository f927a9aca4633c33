cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rms_rate2 -o p --output-format csv -- python3 tools/rms_rate.py > gpurun_out/prof_rms_rate2.log 2>&1; echo "rocprof exit $?"; cat gpurun_out/prof_rms_rate2.log | grep pool
python3 - <<'PY'
import csv, glob
rows=[]
for p in glob.glob('gpurun_out/prof_rms_rate2/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(p)):
        if 'stitch' in row['Kernel_Name']:
            rows.append(int(row['End_Timestamp'])-int(row['Start_Timestamp']))
print('stitch durations (first 400):', sorted(set(rows))[:10], '...', len(rows))
import collections
print(collections.Counter([r//10000 for r in rows]).most_common(12))
PY
