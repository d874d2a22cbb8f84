#!/bin/bash
# config 4 (van der Pol ensemble): kernel stats and matrix-core / VALU counters of the closed-form and the MFMA block solver.
# Counter passes are separate runs (SQ counters, 8 slots per pass); run on the GPU box through gpurun.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_vdp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
for v in valu mfma; do
  FLAG=""; [ $v = mfma ] && FLAG="--mfma"
  ARGS="$GRAFT_REPO_ROOT/bench.py --workload vdp --steps 5 --warmup 1 $FLAG"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$v -o run -- python3 $ARGS > $OUT/bench_$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_$v -o run -- python3 $ARGS > $OUT/bench_pmc_$v.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
res = {}
for v in ("valu", "mfma"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for f in glob.glob(f"{out}/pmc_{v}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            if "vdp_sweep" not in k:
                continue
            agg[k][row["Counter_Name"]] += float(row.get("Counter_Value", 0) or 0)
            if row["Counter_Name"] == "SQ_WAVE_CYCLES":
                cnt[k] += 1
    res[v] = {k: dict(counters={c: x for c, x in d.items()}, dispatches=cnt[k]) for k, d in agg.items()}
    st = {}
    for f in glob.glob(f"{out}/stats_{v}/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "vdp" in row.get("Name", ""):
                st[row["Name"]] = {k2: row[k2] for k2 in ("Calls", "TotalDurationNs", "AverageNs", "Percentage") if k2 in row}
    res[v + "_stats"] = st
json.dump(res, open(f"{out}/vdp_block_solver_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete
grep -i "mfma" $OUT/counters_available.txt | head -20
