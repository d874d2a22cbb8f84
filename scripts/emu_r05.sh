#!/bin/bash
# The time-slice emulation at 1024^3 (scripts/emulate_timeslice.py, spectra on the wire, blocks of 4 sweeps from a predictor) for
# the data flows of sdc_set_timeslice_options (trail sources, last pass: 0 at once / 1 put off + pipelined / 2 put off, split send):
#  * device work per iteration (no message), and with one 8.6 GB device copy standing in for the message's HBM traffic;
#  * wall time per iteration when the message takes 33 ms (8 ranks, two hops) / 67 ms (4 ranks) / 75 ms (2 ranks, host share)
#    to arrive (a wait on the message's own stream): what a time rank's iteration costs end to end.
mkdir -p gpurun_out/r05
run() {  # opts copies msg_ms
  EMU_VARIANTS=spectral EMU_OPTS=$1 EMU_MSG_MS=$3 timeout 900 python scripts/emulate_timeslice.py ${EMU_N:-1024} 8 $2 2> gpurun_out/r05/emu.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())['spectra_on_the_wire']
d['kernels_ms'] = {k: v for k, v in d['kernels_ms'].items()}
print(json.dumps({'options': d['options'], 'message_copies': d['message_copies'], 'message_ms': d['message_ms'], 'message_ms_measured': d.get('message_ms_measured', 0.0), 'ms_per_iteration': round(d['ms_per_iteration'], 2),
                  'device_GB': round(d['device_bytes'] / 1e9, 1), 'kernels_ms_per_iteration': d['kernels_ms_per_iteration']}))
" || tail -3 gpurun_out/r05/emu.err
}
for opts in ${EMU_OPTS_LIST:-0,0,0 0,2,0 0,1,0 5,1,0 5,1,1}; do
  run $opts 0 0
  run $opts 1 0
  for ms in ${EMU_MSG_LIST:-33 67}; do run $opts 1 $ms; done
done | tee gpurun_out/r05/timeslice_emulation.jsonl
