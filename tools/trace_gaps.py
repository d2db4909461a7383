"""Kernel time vs idle gaps of the decode steps from a rocprofv3 --kernel-trace CSV: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# decode steps: between consecutive advance_kernel launches
idx = [i for i, r in enumerate(rows) if "advance_kernel" in r[2]]
if len(idx) < 4:
    sys.exit("no decode steps in the trace")
a, b = idx[-4], idx[-1]                      # three whole steps near the end
seg = rows[a + 1:b + 1]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - rows[a][1]
print(f"{len(seg)} launches in 3 steps: span {span / 3e3:.1f} us/step, kernels {busy / 3e3:.1f} us/step, gaps {(span - busy) / 3e3:.1f} us/step "
      f"({(span - busy) / len(seg) / 1e3:.2f} us per launch)")
agg = {}
for s, e, n in seg:
    k = n.split("(")[0][-60:]
    t = agg.setdefault(k, [0, 0])
    t[0] += 1; t[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t / 3e3:8.1f} us/step  {c // 3:4d} x {t / c / 1e3:7.2f} us  {k}")
