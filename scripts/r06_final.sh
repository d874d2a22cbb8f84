#!/bin/bash
# round 6 evidence run: the driver's bench command, the headline profile (kernel stats + PMC traffic), the config-5 kernel stats and iteration trace
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_final
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
cp gpurun_out/bench_details.json $OUT/bench_default_details.json
bash scripts/profile_bench.sh 1024 > $OUT/profile_bench.log 2>&1
cp gpurun_out/prof_n1024/n1024_kernel_stats.csv gpurun_out/prof_n1024/traffic.json gpurun_out/prof_n1024/pmc_summary.json gpurun_out/prof_n1024/bench_under_rocprof.json $OUT/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ac -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload allencahn --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/ac_bench.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/trace_tail.py $OUT/ac 130 4 > $OUT/allencahn_iteration_trace.txt
cp $(find $OUT/ac -name "*kernel_stats.csv" | head -1) $OUT/allencahn_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
python3 bench.py --workload allencahn --steps 20 --warmup 3 --no-cpu-baseline > $OUT/ac_bench_default.json 2>/dev/null
cat $OUT/bench_default.time; cut -c1-250 $OUT/bench_default.json; cut -c1-250 $OUT/ac_bench_default.json
