#!/bin/bash
# step time of the benchmark configuration over the store threshold of pass 1 (HIBAG_STORE_PAIRS: cells with more pairs are stored
# for pass 2, the others evaluated again) and the fit rule (HIBAG_STORE_FIT); one box, so the lines are comparable
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
for fit in ${FITS:-5 0}; do
for sp in ${PAIRS:-4 6 8 10 12 14 16 20 28 40}; do
  echo -n "STORE_PAIRS=$sp FIT=$fit  "
  HIBAG_STORE_PAIRS=$sp HIBAG_STORE_FIT=$fit timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernels_ms_per_step'], d['roofline']['issue']['cell_sums_stored_per_sample'], d['roofline']['issue']['pairs_evaluated_per_sample']['pass2'])"
done
done
