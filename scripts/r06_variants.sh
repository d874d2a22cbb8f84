#!/bin/bash
# bench prebuilt library variants (build_variants/libsdcmi_*.so, selected through PYSDC_AMD_LIB): kernels of the serial 1024^3 sweep
N=${1:-1024}
cd $GRAFT_REPO_ROOT
for lib in build_variants/libsdcmi_*.so; do
  tag=$(basename $lib .so)
  PYSDC_AMD_LIB=$PWD/$lib python3 bench.py --n $N --steps 4 --warmup 1 --no-cpu-baseline --no-extras --details-file gpurun_out/var_$tag.details.json > gpurun_out/var_$tag.json 2>/dev/null
  python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/var_$tag.details.json"))["headline"]
    print("$tag", round(d["value"],3), "steps/s", {k:round(v["ms_per_launch"],2) for k,v in d["kernels"].items() if v["launches"]>=3 and v["ms_per_launch"] > 3}, 'sweep', round(d['roofline_sweep']['ms_per_sweep'],2))
except Exception as e:
    print("$tag failed", e)
PY
done
