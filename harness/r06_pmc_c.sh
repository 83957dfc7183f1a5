#!/bin/bash
set -u
mkdir -p gpurun_out/r06/final
timeout -k 10 600 python harness/experiments/exp_tail_histogram.py run powerlaw_4m 1.0 256 > gpurun_out/r06/final/experiment_tail_histogram_powerlaw.log 2>&1; echo "tail rc=$?"
grep schedule gpurun_out/r06/final/experiment_tail_histogram_powerlaw.log | cut -c1-400
bash harness/final_measure.sh pmc r06/final papers_like:128:3
ls gpurun_out/.graft_exec_refused 2>/dev/null && tail -3 gpurun_out/.graft_exec_refused
