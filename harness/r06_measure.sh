#!/bin/bash
# How round 6's numbers under profiles/r06/ were produced (each part is one `gpurun -- 'bash harness/r06_measure.sh <part>'` call on a
# fresh one-GPU box; everything lands under gpurun_out/r06/final3/ and is copied / merged into profiles/ afterwards):
#   final      default bench line + `rocprofv3 --kernel-trace --stats` of the same command, PMC passes for the headline, the reddit widths and
#              products-like x 512 (harness/final_measure.sh), the 12-graph evaluation set with the Reorder rows (harness/eval_set.py)
#   headline_eval  bench line + kernel stats + evaluation set only (the round's last tree)
#   big        PMC passes for power-law 4 M x 256 and papers-like x 128, the per-CU finish-time histogram of the power-law configuration
#   lines      bench lines of the other configurations / widths / weighted / backward (after `python harness/merge_traffic.py
#              gpurun_out/r06/final3 r06`, so that they replay the counters), the N > 1 code path on one GPU
#   widths     the remaining widths of the sweep on configurations 3-4 (products-like x 32, power-law x 32 / 128 / 512)
#   suite      smoke() and the whole `pytest -m gpu` suite
#   refresh    cell-by-cell A/B refresh of the shipped tuner buckets for the short-window stand-ins (harness/collect_cells.py)
set -u
PART=${1:-final}
O=gpurun_out/r06/final3; mkdir -p $O
case $PART in
final)
  bash harness/final_measure.sh headline r06/final3
  bash harness/final_measure.sh pmc r06/final3 headline reddit_like:32 reddit_like:512 products_like:512:5
  timeout -k 10 1000 python harness/eval_set.py --reorder --check --output_file $O/results.csv --jsonl $O/eval_set.jsonl > $O/eval_set.log 2>&1; echo "eval rc=$?"
  ;;
headline_eval)
  # the last tree of the round (kernel sources unchanged since the counter passes: same sources hash): bench line + kernel stats + evaluation set
  bash harness/final_measure.sh headline r06/final3
  timeout -k 10 1000 python harness/eval_set.py --reorder --check --output_file $O/results.csv --jsonl $O/eval_set.jsonl > $O/eval_set.log 2>&1; echo "eval rc=$?"
  ;;
big)
  bash harness/final_measure.sh pmc r06/final3 powerlaw_4m:256:3 papers_like:128:3
  timeout -k 10 600 python harness/experiments/exp_tail_histogram.py run powerlaw_4m 1.0 256 > $O/experiment_tail_histogram_powerlaw.log 2>&1; echo "tail rc=$?"
  ;;
lines)
  bash harness/final_measure.sh lines r06/final3
  for ARGS in "--backward" "--weighted --backward" "--weighted --weighted-plane"; do
    NAME=$(echo "$ARGS" | tr -d ' ' | tr -s '-' '_')
    timeout -k 10 400 python bench.py $ARGS --no-cpu-baseline --no-reference-formats > $O/bench_reddit_like_f128${NAME}_final.json 2> $O/bench_reddit$NAME.err
  done
  timeout -k 10 300 python bench.py --gpus 2 --backend gloo --one-device --workload reddit_like --feat 128 --scale 0.25 --config5-scale 0.004 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-formats > $O/bench_gpus2_one_device_dependent_step.json 2> $O/bench_gpus2.err
  timeout -k 10 300 python bench.py --gpus 1 --force-dist --workload reddit_like --feat 128 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-formats > $O/bench_force_dist_1rank_rccl.json 2> $O/bench_force_dist.err
  ;;
widths)
  # SURVEY 8(d) "feat_dim sweep F in {32, 128, 512} on configs 2-4": the widths the `lines` part does not carry
  for WF in products_like:32 powerlaw_4m:32 powerlaw_4m:128 powerlaw_4m:512; do
    W=${WF%%:*}; F=${WF##*:}
    timeout -k 10 500 python bench.py --workload $W --feat $F --steps 5 --warmup 2 --no-cpu-baseline --no-reference-formats > $O/bench_${W}_f${F}_final.json 2> $O/bench_${W}_f$F.err || tail -3 $O/bench_${W}_f$F.err
    python -c "import json; d=json.loads(open('$O/bench_${W}_f${F}_final.json').read().strip().splitlines()[-1]); print('$W', $F, round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4), d['config'].get('tile'))"
  done
  ;;
suite)
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
  timeout -k 10 1150 python -m pytest tests -m gpu -q > $O/pytest_gpu_full_suite.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu_full_suite.log
  ;;
refresh)
  C=gpurun_out/r06/cells; mkdir -p $C
  for G in amazon0505_like amazon0601_like com_amazon_like dd_like ppi_like web_berkstan_like yeast_like yeasth_like; do
    for M in shipped fresh; do
      timeout -k 10 400 python harness/collect_cells.py $M $G $C/${G}_$M.json > $C/${G}_$M.log 2>&1 || tail -3 $C/${G}_$M.log
    done
  done
  python harness/collect_cells.py merge $C $C/merged_store.json
  ;;
esac
