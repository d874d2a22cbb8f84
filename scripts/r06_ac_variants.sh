#!/bin/bash
# config 5 with prebuilt library variants (build_variants/libsdcmi_*.so through PYSDC_AMD_LIB): value without events + kernel table of the second run
cd $GRAFT_REPO_ROOT
for lib in build_variants/libsdcmi_*.so; do
  tag=$(basename $lib .so)
  PYSDC_AMD_LIB=$PWD/$lib python3 bench.py --workload allencahn --steps 20 --warmup 3 --no-cpu-baseline --details-file gpurun_out/acvar_$tag.details.json > gpurun_out/acvar_$tag.json 2>/dev/null
  python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/acvar_$tag.details.json"))["headline"]
    k=d["kernels"]
    print("$tag", round(d["value"],2), "steps/s", round(d["ms_per_step"],3), "ms/step", {n:round(v["ms_per_launch"]*1e3,1) for n,v in k.items() if n.startswith(("fft_y","fft_x"))})
except Exception as e:
    print("$tag failed", e)
PY
done
