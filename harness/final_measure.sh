#!/bin/bash
# The round's measurement batch on the GPU box, part by part (each part fits one gpurun call; raw rocprof output is summarised on
# the box and deleted, gpurun merges at most 64 MiB back):
#   harness/final_measure.sh headline <out>     default bench.py line, rocprofv3 kernel stats of the same command
#   harness/final_measure.sh pmc <out> <workload>...   PMC passes (pmc_bench.sh) + per-step summary (pmc_summarize.py) per workload
#                                                 at F = 128 ("headline" = the default command)
#   harness/final_measure.sh lines <out>        bench lines of the other BASELINE configurations, widths and the weighted path
#   harness/final_measure.sh eval <out>         the 12-graph evaluation set (results.csv, eval_set.jsonl, eval_set.log)
# <out> is a directory under gpurun_out/.  Copy what should be judged into profiles/ afterwards (harness/merge_traffic.py).
set -u
PART=$1; O=gpurun_out/$2; shift 2
cd "$GRAFT_REPO_ROOT" && mkdir -p "$O"
case $PART in
headline)
  python bench.py > $O/bench_reddit_f128_operator.json 2> $O/bench_reddit.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- \
     python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_reddit_f128_operator_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err)
  find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_reddit_f128_operator_kernel_stats.csv \;
  rm -rf $O/stats
  python -c "import json; d=json.load(open('$O/bench_reddit_f128_operator.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms'])"
  ;;
pmc)
  for W in "$@"; do
    # <workload> (F = 128), <workload>:<F>, or <workload>:<F>:<steps> (the big configurations: fewer timed steps per pass)
    WL=${W%%:*}; REST=${W#*:}; F=128; STEPS=""
    if [ "$REST" != "$W" ]; then F=${REST%%:*}; if [ "${REST#*:}" != "$REST" ]; then STEPS="--steps ${REST#*:} --warmup 2"; fi; fi
    if [ "$W" = headline ]; then ARGS="--no-cpu-baseline --no-reference-formats"; else ARGS="--workload $WL --feat $F $STEPS --no-cpu-baseline --no-reference-formats"; fi
    if [ "$W" != headline ]; then W=${WL}_f$F; fi
    python bench.py $ARGS > /dev/null 2>&1      # persists the tile choice: the five passes run the same kernels
    harness/pmc_bench.sh ${O#gpurun_out/}/pmc_$W $ARGS > $O/pmc_$W.txt 2>&1 && python harness/pmc_summarize.py $O/pmc_$W > $O/pmc_${W}_step.txt 2>&1
    rm -rf $O/pmc_$W/pass*/
    tail -1 $O/pmc_${W}_step.txt | cut -c1-300
  done
  ;;
lines)
  for WF in products_like:512 powerlaw_4m:256 papers_like:128 products_like:128 reddit_like:32 reddit_like:512 protein_like:128 yeasth_like:128 ddi_like:128; do
    W=${WF%%:*}; F=${WF##*:}
    timeout -k 10 400 python bench.py --workload $W --feat $F --steps 10 --warmup 3 --no-cpu-baseline --no-reference-formats > $O/bench_${W}_f${F}_final.json 2> $O/bench_${W}_f$F.err
  done
  for W in products_like reddit_like; do
    timeout -k 10 300 python bench.py --weighted --workload $W --feat 128 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-formats > $O/bench_${W}_weighted_f128_final.json 2> $O/bench_${W}_weighted.err
  done
  for f in $O/bench_*_final.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['ms_per_step'],4), 'ms frac', round(d['roofline']['frac'],4), d['roofline'].get('traffic'))" || true; done
  ;;
eval)
  timeout -k 10 1100 python harness/eval_set.py --reorder --check --output_file $O/results.csv --jsonl $O/eval_set.jsonl > $O/eval_set.log 2>&1
  grep -c "" $O/results.csv; grep "Voltrix-fp16" $O/eval_set.log | grep "F=128 " | grep "reorder=N"
  ;;
esac
