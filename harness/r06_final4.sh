#!/bin/bash
set -u
mkdir -p gpurun_out/r06/final3
bash harness/final_measure.sh pmc r06/final3 powerlaw_4m:256:3 papers_like:128:3
bash harness/final_measure.sh lines r06/final3
for ARGS in "--backward" "--weighted --backward" "--weighted --weighted-plane"; do
  NAME=$(echo "$ARGS" | tr -d ' ' | tr -s '-' '_')
  timeout -k 10 400 python bench.py $ARGS --no-cpu-baseline --no-reference-formats > gpurun_out/r06/final3/bench_reddit_like_f128${NAME}_final.json 2> gpurun_out/r06/final3/bench_reddit$NAME.err
  python -c "
import json; d=json.load(open('gpurun_out/r06/final3/bench_reddit_like_f128${NAME}_final.json')); print('$ARGS', round(d['ms_per_step'],4), 'ms', d['roofline']['kernels_ms'])" || tail -3 gpurun_out/r06/final3/bench_reddit$NAME.err
done
ls gpurun_out/.graft_exec_refused 2>/dev/null; true
