"""Per-step GPU timeline from a rocprofv3 kernel trace:  python tools/timeline_summary.py <kernel_trace.csv> <marker substring>
A step starts at every kernel whose name contains the marker; prints the last full step kernel by kernel (start offset, duration,
gap before it) and the averages of (interval, busy) over the last 12 steps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, k in enumerate(ks) if marker in k[2]]
steps = [(a, b) for a, b in zip(starts, starts[1:])]
def short(n):
    n = n.replace("void ", "").replace("pplie::", "")
    return n.split("(")[0][:70]
tail = steps[-12:]
iv = [ (ks[b][0] - ks[a][0]) / 1e3 for a, b in tail]
busy = [sum(k[1] - k[0] for k in ks[a:b]) / 1e3 for a, b in tail]
print("steps seen", len(steps), " last-12 interval us", [round(x) for x in iv], " busy us", [round(x) for x in busy])
for a, b in steps[-3:]:
    t0 = ks[a][0]
    prev = None
    print("---- step, interval %.1f us" % ((ks[b][0] - t0) / 1e3))
    for s, e, n in ks[a:b]:
        gap = 0.0 if prev is None else (s - prev) / 1e3
        print("  +%8.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, short(n)))
        prev = e
