#!/bin/bash
# chunks per work item of the last rounds, re-swept on the round-4 kernels (HIBAG_TAIL_K sets both passes)
cd $GRAFT_REPO_ROOT
for k in "" 1 2 3 4 6 8; do
  for rep in 1 2; do
  HIBAG_TAIL_K=$k timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('K=${k:-default}', round(d['value']), d['roofline']['kernels_ms_per_step'])"
  done
done
