#!/bin/bash
# Kernel-time summary of a short bench run: scripts/prof_quick.sh <tag> [ENV=VAL ...]  -> gpurun_out/<tag>/p_kernel_stats.csv
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/$TAG" -o p -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off > "$ROOT/gpurun_out/$TAG.log" 2>&1
