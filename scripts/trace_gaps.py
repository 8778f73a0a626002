#!/usr/bin/env python3
"""Busy time vs gaps of a rocprofv3 --kernel-trace run of bench.py (small steps: is the chain bound by kernel time, by the
gaps between dependent launches, or by the host?):  python scripts/trace_gaps.py gpurun_out/<dir>/..._kernel_trace.csv [steps]
Takes the last `steps` AdamW launches as step boundaries."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
ends = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
if len(ends) < steps + 1:
    sys.exit(f"only {len(ends)} steps in the trace")
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
seg = ev[lo:hi]
span = seg[-1][1] - seg[0][0]
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
for s, e, _ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
dur = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
    dur[k][0] += e - s; dur[k][1] += 1
gaps = sorted(((seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)), reverse=True)
print(f"{steps} steps: {len(seg) / steps:.1f} launches/step, span {span / steps / 1e3:.1f} us/step, GPU busy {busy / steps / 1e3:.1f} us/step "
      f"({100 * busy / span:.0f} %), sum of kernel durations {sum(d[0] for d in dur.values()) / steps / 1e3:.1f} us/step")
pos = [g for g in gaps if g > 0]
print(f"gaps between consecutive launches: {len(pos) / steps:.1f} per step > 0, mean {sum(pos) / max(len(pos), 1) / 1e3:.2f} us, "
      f"total {sum(pos) / steps / 1e3:.1f} us/step; largest {[round(g / 1e3, 1) for g in gaps[:5]]}")
for k, (t, c) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"  {t / steps / 1e3:8.1f} us/step {c / steps:5.1f} x {t / c / 1e3:7.1f} us  {k}")
