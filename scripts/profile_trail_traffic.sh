#!/bin/bash
# rocprofv3 kernel stats and HBM traffic (FETCH_SIZE / WRITE_SIZE, each in its own pass, kernel trace only) of the time-slice
# emulation with recomputed iterates (EMU_OPTS, default 5,1,0): what the trail's launches really move per dispatch
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_trail
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EMU_VARIANTS=spectral EMU_OPTS=${EMU_OPTS:-5,1,0}
ARGS="$GRAFT_REPO_ROOT/scripts/emulate_timeslice.py ${1:-1024} 4 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ARGS > $OUT/log_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ARGS > $OUT/log_fetch.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ARGS > $OUT/log_write.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/timeslice_kernel_stats.csv
python3 - <<PY
import csv, glob, collections, json
def per_dispatch(kind):
    rows = collections.defaultdict(dict)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % kind, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if not any(s in k for s in ("k_trail", "k_fftx_inv", "k_ffty", "k_fftz_plain")):
                continue
            d = int(row["Dispatch_Id"])
            rows[k][d] = rows[k].get(d, 0.0) + float(row["Counter_Value"] or 0)
    return rows
fetch, write = per_dispatch("fetch"), per_dispatch("write")
out = {}
for k in sorted(set(fetch) | set(write)):
    fv = [v for _, v in sorted(fetch.get(k, {}).items())]
    wv = [v for _, v in sorted(write.get(k, {}).items())]
    m = min(len(fv), len(wv))
    per = [round((2.0 * fv[i] + wv[i]) * 1024.0 / 1e9, 2) for i in range(m)]   # GB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
    out[k[:60]] = {"dispatches": m, "GB_per_dispatch_in_order": per}
json.dump(out, open("$OUT/trail_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, v["dispatches"], v["GB_per_dispatch_in_order"][:24])
PY
find $OUT -name "*.csv" -size +1M -delete
