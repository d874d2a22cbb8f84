#!/bin/bash
# the time-slice emulation at 1024^3 for every data flow of sdc_set_timeslice_options (trail sources, deferred last pass, split send)
mkdir -p gpurun_out/r05
for opts in 0,0,0 5,0,0 0,1,0 5,1,0 5,1,1; do
  for copies in 0 1; do
    EMU_VARIANTS=spectral EMU_OPTS=$opts timeout 900 python scripts/emulate_timeslice.py ${EMU_N:-1024} 8 $copies > gpurun_out/r05/emu_${opts//,/}_c$copies.json 2> gpurun_out/r05/emu_${opts//,/}_c$copies.err || tail -5 gpurun_out/r05/emu_${opts//,/}_c$copies.err
    cat gpurun_out/r05/emu_${opts//,/}_c$copies.json
  done
done
