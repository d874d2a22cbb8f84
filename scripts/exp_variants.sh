#!/bin/bash
# bench every prebuilt library variant under build_variants/ (built in the build container, several at once:
#   hipcc ... -DSDC_<MACRO>=... -o build_variants/lib_<tag>.so pysdc_amd/csrc/sdcmi.hip) on the GPU box:
#   scripts/exp_variants.sh [N] [extra bench args]
N=${1:-1024}; shift
cd $GRAFT_REPO_ROOT
for lib in pysdc_amd/libsdcmi.so build_variants/lib_*.so; do
  tag=$(basename $lib .so)
  PYSDC_AMD_LIB=$PWD/$lib python bench.py --n $N --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/var_$tag.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/var_$tag.json").read().strip().splitlines()[-1])
    print("$tag", round(d["value"],3), "steps/s", {k:round(v["ms_per_launch"],2) for k,v in d["kernels"].items() if v["launches"]>=3 and v["ms_per_launch"] > 5})
except Exception as e:
    print("$tag failed", e)
PY
done
