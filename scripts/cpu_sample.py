"""One bounded sample of the CPU baseline at a size bench.py does not afford inside its own run: the oracle (oracle/sdc_oracle.py,
the NumPy / SciPy restatement of the reference's path) on heat 3-D n^3, M = 5, ONE sweep, CG rtol 1e-12, one core, the
stiffness of the headline workload.  Usage (on the GPU box's host): python scripts/cpu_sample.py 256 > gpurun_out/cpu_sample_256.json
bench.py's cpu_baseline carries the 64^3 and 128^3 samples; this adds the next point of the cost-per-DOF curve."""
import json
import os
import sys

os.environ.setdefault('OMP_NUM_THREADS', '1')
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r = bench._cpu_sample(n, 5, 2.5e-4, 1024, 1)
r['seconds_per_sweep_per_dof'] = r['seconds'] / n**3
r['host_cores'] = os.cpu_count()
print(json.dumps(r))
