"""One train step of a rocprofv3 --kernel-trace CSV as a timeline: python tools/diag/step_timeline.py <kernel_trace.csv> [steps_back | +step_index]
Prints every kernel between the last two optimiser launches (or the pair `steps_back` earlier) in start order: offset from the step's
first kernel, duration, idle time in front of it, overlap marker (the kernel started before the previous one ended: another stream),
short name -- and per-name totals.  (Gaps are inflated under the profiler; durations and order are what this is for.)"""
import csv, sys, collections, re
path, sel = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "0"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
if sel.startswith("+"):      # "+N": the N-th step from the start of the run (0-based)
    a, b = marks[int(sel) - 1] + 1 if int(sel) > 0 else 0, marks[int(sel)] + 1
else:
    a, b = marks[-2 - int(sel)] + 1, marks[-1 - int(sel)] + 1
rows = rows[a:b]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+|void ", "", n)
    return n[:90]


t0, end = rows[0][0], rows[0][0]
tot, cnt = collections.Counter(), collections.Counter()
busy_union = 0
for s, e, n in rows:
    gap = s - end
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {'gap %6.1f' % (gap / 1e3) if gap > 0 else 'overlap   '}  {short(n)}")
    busy_union += max(0, e - max(s, end))
    end = max(end, e)
    tot[short(n)[:60]] += e - s
    cnt[short(n)[:60]] += 1
span = end - t0
print(f"\nkernels {len(rows)}  span {span / 1e6:.3f} ms  busy(union) {busy_union / 1e6:.3f} ms  sum of durations {sum(tot.values()) / 1e6:.3f} ms")
for n, v in tot.most_common(40):
    print(f"  {v / 1e3:9.1f} us  x{cnt[n]:4d}  {n}")
