#!/bin/bash
# Round 6, final measurements on the final kernel sources (the CSR kernel's header changed the sources hash): CSR tests, default bench +
# rocprof stats + PMC for the headline and the reddit widths / products, then the evaluation set.
set -u
O=gpurun_out/r06/final3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csr_path.py tests/test_gpu_stream.py -m gpu -q > $O/pytest_csr.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_csr.log
bash harness/final_measure.sh headline r06/final3
bash harness/final_measure.sh pmc r06/final3 headline reddit_like:32 reddit_like:512 products_like:512:5
timeout -k 10 1000 python harness/eval_set.py --reorder --check --output_file $O/results.csv --jsonl $O/eval_set.jsonl > $O/eval_set.log 2>&1; echo "eval rc=$?"
grep -c "" $O/results.csv; grep "F=128 " $O/eval_set.log | grep "Voltrix " | grep "reorder=N"
