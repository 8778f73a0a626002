#!/bin/bash
# host-side cost of the eager small step (config 1): cProfile of bench.py's step loop, top cumulative entries.
# usage (GPU box): scripts/probe/host_profile.sh [preset] [steps]
P=${1:-config1}; K=${2:-400}
python bench.py --preset $P --steps $K --warmup 50 --no-cpu-baseline --graph off --no-ragged 2>/dev/null | tail -1
python -c "
import cProfile, pstats, sys, runpy
sys.argv = ['bench.py', '--preset', '$P', '--steps', '$K', '--warmup', '50', '--no-cpu-baseline', '--graph', 'off', '--no-ragged']
cProfile.run('runpy.run_path(\"bench.py\", run_name=\"__main__\")', 'gpurun_out/host_prof.out')
" > /dev/null 2>&1
python -c "
import pstats
p = pstats.Stats('gpurun_out/host_prof.out'); p.sort_stats('tottime').print_stats(28)
" | tail -45
