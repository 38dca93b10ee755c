#!/bin/bash
# kernel-by-kernel timeline of the pose-graph LM step at 100k / 400k
set -u
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out/tl; export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/k100 -o pgo -- python $R/tools/pgo_loop.py 100000 400000 3 0 > $R/gpurun_out/tl/loop_100k.log 2>&1
cd $R; grep rep gpurun_out/tl/loop_100k.log | tail -3
f=$(find gpurun_out/tl/k100 -name "*kernel_trace.csv" | head -1)
python tools/timeline_summary.py "$f" pgo_linearize > gpurun_out/tl/timeline_100k.txt 2>&1
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, k in enumerate(ks) if "pgo_linearize" in k[2]]
a, b = starts[-2], starts[-1]
t0 = ks[a][0]; prev = None; busy = 0; gaps = []
from collections import Counter
cnt = Counter(); dur = Counter()
for s, e, n in ks[a:b]:
    n = n.replace("void ", "").replace("pplie::", "").split("(")[0][:50]
    cnt[n] += 1; dur[n] += (e - s) / 1e3; busy += (e - s) / 1e3
    if prev is not None and (s - prev) / 1e3 > 8: gaps.append((round((s - t0) / 1e3), round((s - prev) / 1e3), n))
    prev = e
print("interval us", (ks[b][0] - t0) / 1e3, "busy", round(busy))
for n, d in dur.most_common(12): print("  %-50s x%4d  %8.1f us" % (n, cnt[n], d))
print("gaps > 8 us (at, gap, next kernel):", gaps[:40], "total gap us", sum(g[1] for g in gaps))
PY
rm -f "$f"; find gpurun_out/tl/k100 -name "*.db" -delete
