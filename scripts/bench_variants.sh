cd $GRAFT_REPO_ROOT
for args in "--n 512 --no-spectral-reuse" "--n 512 --skip-residual" "--n 512 --eager-fields" "--n 256 --restol 1e-9" "--workload advdiff --n 256 --skip-residual" "--workload vdp --ntraj 1000000 --mfma" "--n 128 --solver-type CG" "--n 256 --virtual-sweeps 0" "--n 256 --multiplier-table 2 --restol 1e-10" "--n 256 --lazy-predictor-residual" "--n 256 --force-dist" "--n 256 --qi MIN-SR-FLEX" "--n 256 --qi LU --nodes 3 --sweeps 6"; do
  python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-extras --details-file gpurun_out/v.json 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$args', '->', round(d['value'],3), d['unit'], 'finite', d.get('finite'), 'niter', d.get('niter'))
except Exception as e: print('$args', 'FAILED', l[:300])"
done
for args in "--workload allencahn --n 128" "--workload allencahn --n 128 --kernel-events timed" "--workload allencahn --n 64 --ac-variant ref2d --restol 1e-8" "--n 768" "--n 640 --eager-fields" "--n 96"; do
  python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-extras --details-file gpurun_out/v.json 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$args', '->', round(d['value'],3), d['unit'], 'finite', d.get('finite'), 'niter', d.get('niter'))
except Exception as e: print('$args', 'FAILED', l[:300])"
done
