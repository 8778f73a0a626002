#!/bin/bash
# Same-box A/B of two builds of the library: scripts/ab_bench.sh <lib A (or "tree")> <lib B> [rounds] [extra bench args]
# alternates A, B, A, B ... and prints ms/step of every run (boxes of the pool differ by ~1 %: compare within one call).
A=$1; B=$2; R=${3:-2}; shift 3 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
run() {
  if [ "$1" = tree ]; then unset XFMR_HIP_LIB; else export XFMR_HIP_LIB="$ROOT/$1"; fi
  timeout -k 10 300 python "$ROOT/bench.py" --steps 30 --warmup 10 --no-cpu-baseline --no-ragged "${@:2}" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], d['value'], '| resident', d['resident']['ms_per_step'], '| in line: logging', [k['avg_launch_ms'] for k in r['kernels'] if 'logging' in k['kernel']], 'gradient', [k['avg_launch_ms'] for k in r['kernels'] if 'gradient' in k['kernel']], 'ffn fwd / bwd, attn fwd / bwd', [k['avg_launch_ms'] for k in r['kernels'] if k['bound'] == 'hbm'])"
  rc=${PIPESTATUS[0]}; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
}
for i in $(seq 1 $R); do run "$A" "$@"; run "$B" "$@"; done | tee "$ROOT/gpurun_out/ab.log"
