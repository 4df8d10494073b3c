#!/bin/bash
# chunks per work item of pass 1's last rounds (HIBAG_TAIL_K1), interleaved repetitions on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for k in 4 2 3 6 1; do
  HIBAG_TAIL_K1=$k timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('K1=$k', round(d['value']), d['roofline']['kernels_ms_per_step'])"
done
done
