#!/bin/bash
# Engine clock / socket power / temperature while the bench step runs, and while the isolated loss harness runs
# (rocm-smi as an ordinary user; one sample every ~0.7 s).  -> gpurun_out/power_clock.log
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
LOG=gpurun_out/power_clock.log
: > $LOG
sample() {  # $1 = label, $2 = samples
  for i in $(seq 1 $2); do
    echo "[$1 $i] $(rocm-smi -d 0 --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Power|sclk|mclk|Temperature \(Sensor (edge|junction)' | sed 's/GPU\[0\]//; s/ \+/ /g' | tr '\n' ';')" >> $LOG
    sleep 0.5
  done
}
sample idle 3
python bench.py --steps 4000 --warmup 5 --spinup-steps 0 --no-cpu-baseline --graph off > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
sleep 25
sample bench 12
wait $BP
python - <<'PY' >> gpurun_out/power_clock.log
import json
d=json.loads(open("gpurun_out/power_bench.json").read().strip().splitlines()[-1])
print("bench (4000 steps):", d["value"], d["ms_per_step"], "logging in line", d["roofline"]["avg_launch_ms"])
PY
python scripts/bench_logging.py --reps 3000 > gpurun_out/power_harness.log 2>&1 &
HP=$!
sleep 20
sample harness 8
wait $HP
tail -3 gpurun_out/power_harness.log >> $LOG
cat $LOG
