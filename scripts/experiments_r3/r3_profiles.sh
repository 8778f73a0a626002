#!/bin/bash
# kernel traces: config 2 at batch 32 (gap analysis), configs 4 and 5 (one stream) -> gpurun_out/prof_*
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -o p -- python3 $ROOT/bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --overlap off --graph off "$@" > $ROOT/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $ROOT/gpurun_out/prof_$tag.log; return 1; }
}
run b32 --batch 32 && run config4 --preset config4 && run config5 --preset config5 || exit 1
cd $ROOT
for t in b32 config4 config5; do
  f=$(find gpurun_out/prof_$t -name "*kernel_trace.csv" | head -1)
  echo "== $t"; python3 scripts/trace_gaps.py $f 8
done
