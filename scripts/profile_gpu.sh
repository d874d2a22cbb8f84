#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun): kernel stats + HBM traffic counters.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
set -u
N=${1:-1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_n$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --n $N --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ARGS > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -size +8M -delete
python3 - <<PY
import csv, glob, json, collections, os
out = "$OUT"
res = {}
for kind in ("fetch", "write"):
    files = glob.glob(f"{out}/{kind}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            agg[k][0] += float(row.get("Counter_Value", 0) or 0)
            agg[k][1] += 1
    res[kind] = {k: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)} for k, v in agg.items()}
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
print(json.dumps({k: len(v) for k, v in res.items()}))
PY
ls -la $OUT $OUT/stats 2>/dev/null | head -30
