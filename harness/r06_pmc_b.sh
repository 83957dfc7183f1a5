#!/bin/bash
set -u
mkdir -p gpurun_out/r06/final
bash harness/final_measure.sh pmc r06/final powerlaw_4m:256:3
python harness/experiments/exp_tail_histogram.py build > gpurun_out/r06/final/tail_build.log 2>&1
timeout -k 10 600 python harness/experiments/exp_tail_histogram.py run powerlaw_4m 1.0 256 > gpurun_out/r06/final/experiment_tail_histogram_powerlaw.log 2>&1; echo "tail rc=$?"
tail -30 gpurun_out/r06/final/experiment_tail_histogram_powerlaw.log | cut -c1-220
ls gpurun_out/.graft_exec_refused 2>/dev/null && cat gpurun_out/.graft_exec_refused | tail -3
