#!/bin/bash
# SQ counters of the sweep kernels (one pass of 8 SQ counters; kernel-trace only): where do the waves spend their cycles?
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --n ${1:-1024} --steps 2 --warmup 1 --no-cpu-baseline --no-extras --details-file $OUT/bench_details.json"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o run -- python3 $ARGS > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $OUT/p2 -o run -- python3 $ARGS > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(s in k for s in ("k_spec_z", "k_ffty<1024, 8, 1>", "k_fftx_inv<1024, 8, true, false", "k_restrict3", "k_prolong3")):
            continue
        res[k[:40]][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
        if row["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k[:40]] += 1
out = {k: {"dispatches": cnt[k], **{c: v / max(cnt[k], 1) for c, v in d.items()}} for k, d in res.items()}
json.dump(out, open("$OUT/sq_summary.json", "w"), indent=1)
# resident waves per SIMD, MEASURED: SQ_WAVE_CYCLES counts quad-cycles summed over all waves, SQ_BUSY_CYCLES cycles summed over
# the 32 shader engines (8 XCDs x 4) for the time the launch was on the chip; 1024 SIMDs:
#   waves / SIMD = 4 SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024) = SQ_WAVE_CYCLES / (8 SQ_BUSY_CYCLES)
# and the share of the launch's duration a SIMD's vector ALU / a CU's LDS pipe was busy (ACTIVE_INST_* in quad-cycles):
#   VALU: 4 SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES / 32 * 1024),  LDS (one pipe per CU): 4 SQ_ACTIVE_INST_LDS / (SQ_BUSY_CYCLES / 32 * 256)
for k, d in out.items():
    if d.get("SQ_BUSY_CYCLES"):
        d["resident_waves_per_simd"] = d.get("SQ_WAVE_CYCLES", 0.0) / (8.0 * d["SQ_BUSY_CYCLES"])
        d["valu_busy_share_of_simd"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / (8.0 * d["SQ_BUSY_CYCLES"])
        d["lds_busy_share_of_cu_pipe"] = d.get("SQ_ACTIVE_INST_LDS", 0.0) / (2.0 * d["SQ_BUSY_CYCLES"])
json.dump(out, open("$OUT/sq_summary.json", "w"), indent=1)
for k, d in out.items():
    w = d.get("SQ_WAVE_CYCLES", 1)
    print(k, d["dispatches"])
    for c in sorted(d):
        if c != "dispatches":
            print(f"   {c:24s} {d[c]:14.4g}  {d[c] / w:8.3f} of wave cycles")
PY
find $OUT -name "*.csv" -size +1M -delete
