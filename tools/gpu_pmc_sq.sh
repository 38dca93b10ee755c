#!/bin/bash
# SQ counters (instruction mix / stall buckets) of every kernel a command launches: one rocprofv3 --pmc pass, kernel-trace only.
#   tools/gpu_pmc_sq.sh <tag> <kernel-substring> -- <command...>
set -u
TAG=$1; PAT=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/pmc_sq
cd /tmp && export TMPDIR=/tmp
CTRS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/$TAG -o pmc -- "$@" > $R/gpurun_out/pmc_sq/$TAG.log 2>&1
tail -2 $R/gpurun_out/pmc_sq/$TAG.log
cd $R
python - "$TAG" "$PAT" <<'PY'
import csv, glob, sys, collections, json
tag, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_sq/{tag}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = m.get("SQ_WAVES", 0) or 1
    m["valu_per_wave"] = m.get("SQ_INSTS_VALU", 0) / w
    m["salu_per_wave"] = m.get("SQ_INSTS_SALU", 0) / w
    m["launches"] = len(next(iter(d.values())))
    out[k] = m
print(json.dumps(out, indent=1))
json.dump(out, open(f"gpurun_out/pmc_sq/{tag}.json", "w"), indent=1)
PY
find gpurun_out/pmc_sq/$TAG -name "*.csv" -size +2M -delete
