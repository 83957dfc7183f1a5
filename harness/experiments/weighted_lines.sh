#!/bin/bash
# Round 6: the weighted tests and the weighted bench lines (papers-like: the per-call rule takes the value plane; reddit / products-like: the
# separable path).  Output under gpurun_out/r06/final3/.
set -u
O=gpurun_out/r06/final3; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_weighted.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 700 python bench.py --weighted --workload papers_like --feat 128 --steps 5 --warmup 2 --no-cpu-baseline --no-reference-formats > $O/bench_papers_like_weighted_f128_final.json 2> $O/bench_papers_weighted.err
python -c "
import json; d=json.load(open('$O/bench_papers_like_weighted_f128_final.json')); print('papers_like', round(d['ms_per_step'],3), d['roofline']['kernels_ms'], d['config']['rowsum_check_max_rel_err'], str(d['config']['sparse_format'])[:200])" || tail -5 $O/bench_papers_weighted.err
for W in reddit_like products_like; do
  timeout -k 10 300 python bench.py --weighted --workload $W --feat 128 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-formats > $O/bench_${W}_weighted_f128_final.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_${W}_weighted_f128_final.json')); print('$W', round(d['ms_per_step'],3), d['roofline']['kernels_ms'])"
done
