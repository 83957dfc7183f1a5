#!/bin/bash
# Run bench.py (default tune space: the first call sweeps tile x schedule) on the BASELINE stand-ins and bring the persisted
# choices (exact keys + graph-statistics buckets, voltrix/jit_kernels/tuner.py) back under gpurun_out/: the bucket entries are
# what voltrix/jit_kernels/tuned_defaults.json ships, so that a new graph of a known shape class starts without a sweep.
#   usage (GPU box): harness/collect_tuned.sh <outdir under gpurun_out> [workload:feat ...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for WF in "$@"; do
  W=${WF%%:*}; F=${WF##*:}
  timeout -k 10 500 python3 bench.py --workload $W --feat $F --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${W}_f$F.json 2> $OUT/bench_${W}_f$F.err
  echo "$W F=$F exit=$? $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_${W}_f$F.json')); print(round(d['ms_per_step'],4), 'ms', d['config']['tile'], d['config'].get('first_call_ms'))" 2>/dev/null)"
done
cp voltrix-spmm_amd/.jit_cache/tuned.json $OUT/tuned.json 2>/dev/null
