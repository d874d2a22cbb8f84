cd $GRAFT_REPO_ROOT
for lib in build_variants/libsdcmi_q0.so build_variants/libsdcmi_q3.so; do
  for args in "--n 512 --no-spectral-reuse" "--n 1024 --no-spectral-reuse --steps 2"; do
  PYSDC_AMD_LIB=$PWD/$lib python3 bench.py $args --warmup 1 --no-cpu-baseline --no-extras --details-file gpurun_out/qv.json > /dev/null 2>&1
  python3 - <<PY
import json
d=json.load(open("gpurun_out/qv.json"))["headline"]
print("$lib", "$args", round(d["value"],3), {n:round(v["ms_per_launch"],3) for n,v in d["kernels"].items() if n.split('[')[0] in ("gather","residual","end_point","integrate","stencil")})
PY
  done
done
