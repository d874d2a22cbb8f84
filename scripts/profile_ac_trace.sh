#!/bin/bash
# kernel trace of the two-level Allen-Cahn bench: the launches of the last iterations with their gaps (scripts/trace_tail.py)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ac_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload allencahn --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
python3 $GRAFT_REPO_ROOT/scripts/trace_tail.py $OUT/stats ${1:-230} ${2:-12} > $OUT/tail.txt
python3 $GRAFT_REPO_ROOT/scripts/trace_gaps.py $OUT/stats 0.5 > $OUT/gaps.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/gaps.txt
