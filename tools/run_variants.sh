#!/bin/bash
# runs bench.py against every gpurun_var_*.so (tuning experiments; built locally with -DHIBAG_TILE / -DHIBAG_CHUNK)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
for so in gpurun_var_*.so; do
  HIBAG_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$so', round(d['value']), d['roofline']['kernels_ms_per_step'])"
done
