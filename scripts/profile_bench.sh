#!/bin/bash
# rocprofv3 evidence for the headline bench (run on the GPU box through gpurun): kernel stats of the same command the
# bench line comes from, then HBM traffic counters in their OWN passes (FETCH_SIZE and WRITE_SIZE do not fit one pass
# on gfx950; never combined with tracing other than --kernel-trace).  Writes gpurun_out/prof_n<N>/{*_kernel_stats.csv,
# pmc_summary.json, traffic.json}; copy what is to be judged into profiles/<round>/.
set -u
N=${1:-1024}
EXTRA=${2:-}   # further bench.py flags (e.g. "--eager-fields", "--restol 1e-10 --steps 1"), TAG names the output directory
TAG=${3:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_n$N$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --n $N --steps 3 --warmup 1 --no-cpu-baseline --no-extras --details-file $OUT/bench_details_$$.json $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ARGS > $OUT/bench_write.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/make_traffic.py $OUT $N
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/n${N}_kernel_stats.csv
grep "^{" $OUT/bench_stats.log | tail -1 > $OUT/bench_under_rocprof.json
ls -la $OUT
