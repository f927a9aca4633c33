set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/rms_rate.py > gpurun_out/rms_rate.log 2>&1; echo "rms_rate exit $?"; cat gpurun_out/rms_rate.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rms or epilogue or decorrelate or chain" > gpurun_out/pytest_r2_rms.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_r2_rms.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rms_rate -o p --output-format csv -- python3 tools/rms_rate.py > gpurun_out/prof_rms_rate.log 2>&1; echo "rocprof exit $?"
python3 - <<'PY'
import csv, glob
for p in glob.glob('gpurun_out/prof_rms_rate/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(p)):
        if 'vnd' in row['Name']: print(row['Name'][:70], row['Calls'], row['AverageNs'], row['MinNs'])
PY
