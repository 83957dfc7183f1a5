#!/bin/bash
# Round 6, GPU batch 1: the new tests, the default bench line (with the rocSPARSE cells), the weighted / backward lines, the CU-mask
# experiment.  Everything writes under gpurun_out/r06/.
set -u
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_cluster_order.py tests/test_gpu_weighted.py tests/test_gpu_stream.py tests/test_gpu_schedule.py tests/test_gpu_autograd.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -3 $O/pytest_new.log
python bench.py > $O/bench_reddit_f128_operator.json 2> $O/bench_reddit.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_reddit_f128_operator.json"))
print("headline", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["gather_ceiling"], d["config"].get("fp32_in_ms_per_step"))
print(json.dumps(d["vendor_gpu_baseline"])[:1500])
PY
for ARGS in "--weighted" "--weighted --weighted-plane" "--backward" "--weighted --backward"; do
  NAME=$(echo "$ARGS" | tr -d ' ' | tr -s '-' '_')
  timeout -k 10 400 python bench.py $ARGS --no-cpu-baseline --no-reference-formats > $O/bench_reddit_f128$NAME.json 2> $O/bench_reddit$NAME.err; echo "bench $ARGS rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_reddit_f128$NAME.json')); print('$ARGS', round(d['ms_per_step'],4), 'ms', d['roofline']['kernels_ms'], d['config']['sparse_format'].get('values','')[:60], d['config']['rowsum_check_max_rel_err'])" || tail -5 $O/bench_reddit$NAME.err
done
timeout -k 10 300 python harness/experiments/exp_cu_mask.py > $O/experiment_cu_mask.json 2> $O/experiment_cu_mask.err; echo "cu mask rc=$?"; cut -c1-1500 $O/experiment_cu_mask.json; tail -3 $O/experiment_cu_mask.err
