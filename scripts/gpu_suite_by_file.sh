#!/bin/bash
# the GPU suite file by file (a GPU memory fault ends one file's run, not the rest); summary lines in gpurun_out/suite_by_file.txt
# usage: [SDC_LAZY_MIN_BYTES=4096] scripts/gpu_suite_by_file.sh [pytest args]
mkdir -p gpurun_out/suite
: > gpurun_out/suite_by_file.txt
for f in tests/test_gpu_*.py; do
  b=$(basename $f .py)
  timeout 3000 python -m pytest $f -q -m gpu -x "$@" > gpurun_out/suite/$b.txt 2>&1
  rc=$?
  echo "$b rc=$rc $(grep -E 'passed|failed|error' gpurun_out/suite/$b.txt | tail -1)" >> gpurun_out/suite_by_file.txt
  if [ $rc -ne 0 ]; then grep -v "^  File\|^Extension modules" gpurun_out/suite/$b.txt | tail -25 >> gpurun_out/suite_by_file.txt; fi
done
cat gpurun_out/suite_by_file.txt
