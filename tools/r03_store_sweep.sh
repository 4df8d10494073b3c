#!/bin/bash
# round 3: which cell sums pass 1 stores -- the threshold (HIBAG_STORE_PAIRS) and the "fit the visit into one block" rule
# (HIBAG_STORE_FIT = smallest cell it may store, 0 = off) -- against the block-stream pass 2.  Arguments: pairs "P:F" ...
cd $GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- 12:0 12:3 12:5 12:8 16:5 8:5 24:5 12:2
for cfg in "$@"; do
  sp=${cfg%%:*}; sf=${cfg##*:}
  HIBAG_STORE_PAIRS=$sp HIBAG_STORE_FIT=$sf HIBAG_DEBUG_MODEL=1 timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('store>$sp fit>=$sf', round(d['value']), d['roofline']['kernels_ms_per_step'], d['roofline']['issue']['cell_sums_stored_per_sample'], d['roofline']['issue']['pairs_evaluated_per_sample']['pass2'])"
  grep "hibag model" /tmp/err.txt | head -1 | sed 's/.*blocks of 32/   blocks of 32/; s/; pair lists.*//'
done
