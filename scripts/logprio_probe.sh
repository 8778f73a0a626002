python -c "
import torch
s=torch.cuda.Stream(priority=1); print('prio1', s.priority); s=torch.cuda.Stream(priority=-1); print('prio-1', s.priority); s=torch.cuda.Stream(); print('default', s.priority)"
for i in 1 2; do
for v in 0 low; do XFMR_LOG_STREAM_PRIORITY=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LOGPRIO=$v', d['ms_per_step'], round(d['value']))"; done; done
