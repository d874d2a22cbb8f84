#!/bin/bash
# SQ counters of the z launches of the time-slice emulation (trail data flow, split send: one launch per number of replayed sweeps)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sq_trail
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EMU_VARIANTS=spectral EMU_OPTS=${EMU_OPTS:-5,1,1}
ARGS="$GRAFT_REPO_ROOT/scripts/emulate_timeslice.py ${1:-1024} 4 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o run -- python3 $ARGS > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $OUT/p2 -o run -- python3 $ARGS > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_spec_z" not in k and "k_trail" not in k:
            continue
        res[k[:44]][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
        if row["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_WAVES"):
            cnt[(k[:44], row["Counter_Name"])] += 1
out = {}
for k, d in res.items():
    n1, n2 = max(cnt[(k, "SQ_WAVE_CYCLES")], 1), max(cnt[(k, "SQ_WAVES")], 1)
    out[k] = {c: v / (n2 if c in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VMEM", "SQ_WAVES", "SQ_INSTS_SMEM") else n1) for c, v in d.items()}
    out[k]["dispatches"] = n1
json.dump(out, open("$OUT/sq_summary.json", "w"), indent=1)
for k, d in out.items():
    w = d.get("SQ_WAVE_CYCLES", 1)
    waves = d.get("SQ_WAVES", 1)
    print(k, d["dispatches"])
    for c in sorted(d):
        if c != "dispatches":
            print(f"   {c:24s} {d[c]:14.4g}  {d[c] / w:8.3f} of wave cycles  {d[c] / waves:10.1f} per wave")
PY
find $OUT -name "*.csv" -size +1M -delete
