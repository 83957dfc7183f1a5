#!/bin/bash
# Round 6: refresh the shipped buckets of the short-window stand-ins cell by cell (harness/collect_cells.py), A/B on this box.
set -u
O=gpurun_out/r06/cells; mkdir -p $O
for G in amazon0505_like amazon0601_like com_amazon_like dd_like ppi_like web_berkstan_like yeast_like yeasth_like; do
  for M in shipped fresh; do
    timeout -k 10 400 python harness/collect_cells.py $M $G $O/${G}_$M.json > $O/${G}_$M.log 2>&1 || tail -3 $O/${G}_$M.log
  done
done
python harness/collect_cells.py merge $O $O/merged_store.json
