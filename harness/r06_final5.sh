#!/bin/bash
set -u
mkdir -p gpurun_out/r06/final3
bash harness/final_measure.sh lines r06/final3
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/r06/final3/pytest_gpu_full_suite.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06/final3/pytest_gpu_full_suite.log
