# rocprofv3 kernel-trace stats of the default bench line -> gpurun_out/prof_one (summarise with scripts/rocpd_stats.py)
set -eo pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_one -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --overlap off > $ROOT/gpurun_out/prof_one.log 2>&1
