#!/bin/bash
# rebuild the library on the GPU box with experiment macros and run a short bench:
#   scripts/exp_build.sh TAG N [-D...]        (BENCH_ARGS="--skip-residual" adds bench flags)
TAG=$1; N=$2; shift; shift
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o pysdc_amd/libsdcmi.so pysdc_amd/csrc/sdcmi.hip -lrt 2>/dev/null || { echo "build failed $TAG"; exit 1; }
python bench.py --n $N --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-extras --details-file gpurun_out/exp_$TAG.details.json ${BENCH_ARGS:-} > gpurun_out/exp_$TAG.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/exp_$TAG.details.json"))["headline"]
print("$TAG", round(d["value"],3), "steps/s", round(d["ms_per_step"],2), "ms/step", {k:round(v["ms_per_launch"],3) for k,v in d["kernels"].items() if v["launches"]>=4}, 'sweep', round(d['roofline_sweep']['ms_per_sweep'],2))
PY
