#!/bin/bash
set -u
O=gpurun_out/r06/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_full_suite.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu_full_suite.log
