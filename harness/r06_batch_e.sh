#!/bin/bash
# Round 6, GPU batch E: held-out tuner test after the sweep change, the 2-rank one-device line (gloo: two ranks cannot share one
# device under RCCL), then the final 12-graph evaluation set.
set -u
O=gpurun_out/r06/final; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tuner_bucket.py -m gpu -q -s -k "held_out" > $O/pytest_heldout.log 2>&1; echo "pytest rc=$?"; grep "held-out\|passed\|failed" $O/pytest_heldout.log | cut -c1-420
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --one-device --workload reddit_like --feat 128 --scale 0.25 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-formats > $O/bench_gpus2_one_device_dependent_step.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?"
python -c "
import json; d=json.load(open('$O/bench_gpus2_one_device_dependent_step.json')); print(d['n_gpus'], round(d['ms_per_step'],4), d['config'].get('allgather_ms'), d['config'].get('local_spmm_ms'), d['config']['parallelism'][:100])" || tail -5 $O/bench_gpus2.err
timeout -k 10 1000 python harness/eval_set.py --reorder --check --output_file $O/results.csv --jsonl $O/eval_set.jsonl > $O/eval_set.log 2>&1; echo "eval rc=$?"
grep -c "" $O/results.csv; grep "F=128 " $O/eval_set.log | grep "Voltrix-fp16"
