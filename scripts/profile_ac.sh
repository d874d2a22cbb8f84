#!/bin/bash
# rocprofv3 kernel statistics of the two-level Allen-Cahn bench (run on the GPU box through gpurun)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload allencahn --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +8M -delete
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/allencahn_kernel_stats.csv
head -16 "$f" | cut -c1-110
