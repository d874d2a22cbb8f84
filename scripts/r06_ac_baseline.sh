#!/bin/bash
# round 6 baseline of config 5: bench with and without HIP events around the launches, kernel trace without events
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_ac0
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --workload allencahn --steps 20 --warmup 3 --no-cpu-baseline > $OUT/ev_on.json 2>$OUT/ev_on.err
python3 bench.py --workload allencahn --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events > $OUT/ev_off.json 2>$OUT/ev_off.err
cut -c1-300 $OUT/ev_on.json; cut -c1-300 $OUT/ev_off.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload allencahn --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/bench.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/trace_tail.py $OUT/stats ${1:-200} ${2:-12} > $OUT/tail.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
tail -3 $OUT/tail.txt
