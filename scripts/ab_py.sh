#!/bin/bash
# A/B of a Python-side change: scripts/ab_py.sh <file in tree> <alternative copy> [rounds]; alternates the two versions
F=$1; ALT=$2; R=${3:-3}
cp "$F" /tmp/ab_cur.py
for i in $(seq 1 $R); do
  for v in cur alt; do
    if [ $v = cur ]; then cp /tmp/ab_cur.py "$F"; else cp "$ALT" "$F"; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], round(d['value']))"
  done
done
cp /tmp/ab_cur.py "$F"
