set -eo pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for d in 0 1 3 7 8; do
export XFMR_ATTN_DBG=$d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/attn_dbg_$d -o bench -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --overlap off > $ROOT/gpurun_out/attn_dbg_$d.log 2>&1
done
