#!/bin/bash
# round 6: config-5 correctness + speed after a kernel change: nested-transfer tests, multi-level tests, the bench without events
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_ac1
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_transfer_nested.py tests/test_gpu_multilevel.py tests/test_gpu_configs.py tests/test_gpu_plugin.py -x -q -m gpu 2>&1 | tail -15 > $OUT/tests.txt
cat $OUT/tests.txt
python3 bench.py --workload allencahn --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events > $OUT/ev_off.json 2>$OUT/ev_off.err
cut -c1-300 $OUT/ev_off.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload allencahn --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/bench.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/trace_tail.py $OUT/stats ${1:-160} ${2:-12} > $OUT/tail.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
tail -3 $OUT/tail.txt
