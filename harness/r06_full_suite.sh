#!/bin/bash
set -u
O=gpurun_out/r06/final; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -q > $O/pytest_gpu_full_suite.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu_full_suite.log
