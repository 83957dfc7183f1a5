#!/bin/bash
# Round 6, GPU batch D: the N > 1 code path on one GPU (artefacts), final bench lines of the BASELINE configurations and widths with
# the replayed counters, the weighted / backward lines, the new tests again on the final tree.
set -u
O=gpurun_out/r06/final; mkdir -p $O
timeout -k 10 300 python bench.py --gpus 2 --one-device --workload reddit_like --feat 128 --scale 0.25 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-formats > $O/bench_gpus2_one_device_dependent_step.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --force-dist --workload reddit_like --feat 128 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-formats > $O/bench_force_dist_1rank_rccl.json 2> $O/bench_force_dist.err; echo "force-dist rc=$?"
for f in $O/bench_gpus2_one_device_dependent_step.json $O/bench_force_dist_1rank_rccl.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], d['n_gpus'], round(d['ms_per_step'],4), d['config'].get('allgather_ms'), d['config'].get('local_spmm_ms'), d['config']['parallelism'][:80])" || tail -3 ${f%.json}.err; done
bash harness/final_measure.sh lines r06/final
for ARGS in "--backward" "--weighted --backward" "--weighted --weighted-plane"; do
  NAME=$(echo "$ARGS" | tr -d ' ' | tr -s '-' '_')
  timeout -k 10 400 python bench.py $ARGS --no-cpu-baseline --no-reference-formats > $O/bench_reddit_like_f128${NAME}_final.json 2> $O/bench_reddit$NAME.err
  python -c "
import json; d=json.load(open('$O/bench_reddit_like_f128${NAME}_final.json')); print('$ARGS', round(d['ms_per_step'],4), 'ms', d['roofline']['kernels_ms'])" || tail -3 $O/bench_reddit$NAME.err
done
timeout -k 10 900 python -m pytest tests/test_cluster_order.py tests/test_gpu_weighted.py tests/test_gpu_tuner_bucket.py tests/test_gpu_full_size.py -m gpu -q -k "cluster or weighted or held_out or backward or separable or scale_rows" > $O/pytest_new_final.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_new_final.log; grep "held-out\|forward " $O/pytest_new_final.log
