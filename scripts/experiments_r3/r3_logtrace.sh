#!/bin/bash
# the logging pass three ways on ONE box: isolated harness, in line in the bench (HIP events), one-stream rocprofv3 trace
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/logtrace
export PYTHONUNBUFFERED=1
python scripts/bench_logging.py --reps 40 2>&1 | tail -4
timeout -k 10 300 python bench.py --no-cpu-baseline --graph off > gpurun_out/logtrace/bench.json 2> gpurun_out/logtrace/bench.err || { tail gpurun_out/logtrace/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/logtrace/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("bench value", d["value"], d["ms_per_step"], "| logging in line", r["avg_launch_ms"], r["frac"], "| gradient", r["kernels"][0]["avg_launch_ms"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/logtrace/trace
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/logtrace/trace -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --overlap off --graph off > $GRAFT_REPO_ROOT/gpurun_out/logtrace/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,statistics
f=glob.glob('gpurun_out/logtrace/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for key in ['loss_main_dma_kernel<128, -3>','loss_main_dma_kernel<128, 7>']:
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if key in r['Kernel_Name']]
    print('trace', key, len(d), 'min', min(d), 'median', statistics.median(d), 'mean', sum(d)/len(d))
PY
