#!/bin/bash
set -u
mkdir -p gpurun_out/r06/final
: > gpurun_out/.graft_exec_refused 2>/dev/null || true
bash harness/final_measure.sh headline r06/final
bash harness/final_measure.sh pmc r06/final headline reddit_like:32 reddit_like:512 products_like:512:5
echo "exec refused lines: $(wc -l < gpurun_out/.graft_exec_refused 2>/dev/null || echo none)"
