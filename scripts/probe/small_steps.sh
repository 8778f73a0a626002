#!/bin/bash
# the small-step shapes, one bench line each: scripts/probe/small_steps.sh <tag>  -> gpurun_out/<tag>/*.json + a summary
TAG=${1:-small}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() { name=$1; shift; timeout -k 10 240 python bench.py --no-cpu-baseline --no-ragged "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "FAILED $name"; }
run refdefault --preset reference-default
run b32 --batch 32
run b128 --batch 128
run config1 --preset config1
run config4 --preset config4
[ "$2" = "nohead" ] || run b512
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["config"].get("launch"), d.get("graph_autotune_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
