for i in 1 2; do
for v in 0 1; do XFMR_DW_SIDE=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DW_SIDE=$v', d['ms_per_step'], round(d['value']))"; done; done
