"""GPU idle gaps of a run from a rocprofv3 --kernel-trace CSV: python tools/diag/gap_report.py <kernel_trace.csv> [min_gap_us] [tail_fraction]
Sorts the kernels of the LAST tail_fraction of the run by start time, sums the idle time between consecutive kernels and lists the largest
gaps with the kernels on both sides (who was the GPU waiting for: the host, or a dependent launch)."""
import csv, sys, collections
path, min_gap, tail = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 15.0, float(sys.argv[3]) if len(sys.argv) > 3 else 0.35
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
if len(marks) >= 5:        # the last four train steps: from one optimiser launch to the one four steps later
    rows = rows[marks[-5] + 1:marks[-1] + 1]
    print("window: 4 steps between adamw launches")
else:
    t0, t1 = rows[0][0], rows[-1][1]
    rows = [r for r in rows if r[0] >= t1 - (t1 - t0) * tail]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
end = rows[0][1]; prev = rows[0][2]
for s, e, n in rows[1:]:
    if s > end:
        gaps.append((s - end, prev, n))
    if e > end:
        end, prev = e, n
tot_gap = sum(g for g, _, _ in gaps)
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy(sum) {busy / 1e6:.2f} ms  idle {tot_gap / 1e6:.2f} ms ({100 * tot_gap / span:.1f} %)")
small = sum(g for g, _, _ in gaps if g < min_gap * 1e3)
print(f"gaps < {min_gap} us: {sum(1 for g, _, _ in gaps if g < min_gap * 1e3)} totalling {small / 1e6:.2f} ms; >= : {sum(1 for g, _, _ in gaps if g >= min_gap * 1e3)} totalling {(tot_gap - small) / 1e6:.2f} ms")
by = collections.Counter()
for g, a, b in gaps:
    if g >= min_gap * 1e3:
        by[(a[:48], b[:48])] += g
for (a, b), g in by.most_common(25):
    print(f"  {g / 1e3:9.1f} us  after {a:48s} before {b}")
