#!/bin/bash
set -u
O=gpurun_out/r06/final2; mkdir -p $O
python bench.py > $O/bench_reddit_f128_operator.json 2> $O/bench_reddit.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_reddit_f128_operator.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['fp32_in_ms_per_step'], d['vendor_gpu_baseline']['rocsparse']['fp16']['rocsparse_best_ms'])"
timeout -k 10 1100 python harness/eval_set.py --reorder --check --output_file $O/results.csv --jsonl $O/eval_set.jsonl > $O/eval_set.log 2>&1; echo "eval rc=$?"
grep -c "" $O/results.csv; grep "F=128 " $O/eval_set.log | grep "Voltrix-fp16"
