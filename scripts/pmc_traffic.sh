#!/bin/bash
# Memory-side traffic of EVERY kernel of the training step: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots),
# L2 hit / miss in a third.  scripts/pmc_traffic.sh <tag> -> gpurun_out/<tag>/pmc_fetch|pmc_write|pmc_l2
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export XFMR_DW_SIDE=0  # one stream: counters per kernel, not per overlap
B="--steps 2 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o step -- python3 "$ROOT/bench.py" $B > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o step -- python3 "$ROOT/bench.py" $B > "$OUT/pmc_write.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_l2" -o step -- python3 "$ROOT/bench.py" $B > "$OUT/pmc_l2.log" 2>&1
echo done > "$OUT/PMC_TRAFFIC_DONE"
