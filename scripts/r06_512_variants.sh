#!/bin/bash
# configs 2 and 3 (512^3) with prebuilt library variants
cd $GRAFT_REPO_ROOT
for lib in build_variants/libsdcmi_*.so; do
  tag=$(basename $lib .so)
  for wl in "heat" "advdiff"; do
  PYSDC_AMD_LIB=$PWD/$lib python3 bench.py --workload $wl --n 512 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --details-file gpurun_out/v512.json > /dev/null 2>&1
  python3 - <<PY
import json
d=json.load(open("gpurun_out/v512.json"))["headline"]
print("$tag", "$wl", round(d["value"],2), {n:round(v["ms_per_launch"],3) for n,v in d["kernels"].items() if v["launches"]>=10 and v["ms_per_launch"]>0.5})
PY
  done
done
