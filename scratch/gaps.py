"""GPU idle time inside one step of a rocprofv3 --kernel-trace: python scratch/gaps.py <runc_kernel_trace.csv> [step index from the end, default 1]
A step runs from one head_flags_kernel to the next; busy = union of the dispatch intervals."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-70:]) for r in rows)
starts = [i for i, e in enumerate(ev) if 'head_flags_kernel' in e[2]]
a, b = starts[-1 - back], starts[-back]
step = ev[a:b]
t0 = step[0][0]; t1 = max(e[1] for e in step)
busy = 0; cs, ce = step[0][0], step[0][1]; gaps = []; prev = step[0][2]
for s, e, n in step[1:]:
    if s > ce: busy += ce - cs; gaps.append((s - ce, prev, n)); cs, ce = s, e
    else: ce = max(ce, e)
    prev = n
busy += ce - cs
print("step span %.3f ms, busy %.3f ms, idle %.3f ms in %d gaps (%d dispatches)" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), len(step)))
for g, p, n in sorted(gaps, reverse=True)[:30]: print("  %6.1f us  after %-50s before %s" % (g / 1e3, p[-50:], n[-50:]))
