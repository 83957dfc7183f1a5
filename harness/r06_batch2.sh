#!/bin/bash
# Round 6, GPU batch 2: stream-kernel tests again (asm stores + hazard nop), reorder / tuner tests, the value-plane weighted line, then
# the 12-graph evaluation set with rocSPARSE-best and the Reorder rows from shuffled labels.
set -u
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py tests/test_reorder.py tests/test_gpu_reorder_search.py tests/test_gpu_tuner_bucket.py -m gpu -q > $O/pytest_batch2.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_batch2.log; grep "held-out" $O/pytest_batch2.log
timeout -k 10 400 python bench.py --weighted --weighted-plane --no-cpu-baseline --no-reference-formats > $O/bench_reddit_f128_weighted_plane.json 2> $O/bench_reddit_weighted_plane.err; echo "bench plane rc=$?"
python -c "
import json; d=json.load(open('$O/bench_reddit_f128_weighted_plane.json')); print('weighted plane', round(d['ms_per_step'],4), 'ms', d['roofline']['kernels_ms'])"
timeout -k 10 1000 python harness/eval_set.py --reorder --check --output_file $O/results.csv --jsonl $O/eval_set.jsonl > $O/eval_set.log 2>&1; echo "eval rc=$?"
grep -c "" $O/results.csv; grep "F=128 " $O/eval_set.log | grep "Voltrix-fp16\|rocSPARSE-best-fp16\|CSR-gather-fp16"
