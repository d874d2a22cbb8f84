#!/bin/bash
# quick look at the trail data flows of the time-slice emulation: EMU_N (default 1024)
mkdir -p gpurun_out/r05
for opts in ${EMU_OPTS_LIST:-5,1,0 5,1,1}; do
  EMU_VARIANTS=spectral EMU_OPTS=$opts timeout 900 python scripts/emulate_timeslice.py ${EMU_N:-1024} 8 0 2> gpurun_out/r05/emu_q.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())['spectra_on_the_wire']
print(d['options'], 'ms/iteration %.2f' % d['ms_per_iteration'], 'GB %.1f' % (d['device_bytes'] / 1e9), {k: v for k, v in d['kernels_ms'].items()})
"
done
