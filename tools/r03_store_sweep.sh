#!/bin/bash
# round 3: the store threshold (pairs per cell above which pass 1 stores the cell's sum) against the block-stream pass 2
cd $GRAFT_REPO_ROOT
for sp in 4 6 8 10 12 16 24 40; do
  HIBAG_STORE_PAIRS=$sp HIBAG_DEBUG_MODEL=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('store>$sp', round(d['value']), d['roofline']['kernels_ms_per_step'], d['roofline']['issue']['cell_sums_stored_per_sample'], d['roofline']['issue']['pairs_evaluated_per_sample']['pass2'])"
  grep "hibag model" /tmp/err.txt | head -1 | sed 's/.*blocks of 32/blocks of 32/'
done
