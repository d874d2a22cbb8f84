#!/bin/bash
# LDS activity / bank conflicts per kernel (rocprofv3 PMC pass of a short bench run): scripts/profile_lds.sh [N]
set -u
N=${1:-1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lds_n$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -o run -- python3 $GRAFT_REPO_ROOT/bench.py --n $N --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("$OUT/lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:10]:
    g = max(v.get("GRBM_GUI_ACTIVE", 1), 1)
    print(f"{k:62s} n={cnt[k]:3d} lds_active/cycle/CU={v.get('SQ_LDS_IDX_ACTIVE',0)/g/256:.3f} bank_conflict/active={v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1):.3f}")
PY
