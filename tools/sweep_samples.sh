#!/bin/bash
# bench.py over a list of batch sizes (how much of a step is launch quantisation): tools/sweep_samples.sh 8192 10000 ...
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --samples $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels_ms_per_step']; print($n, round(d['value']), k, 'us per 1k samples: total %.1f accum %.1f' % (1e6*k['total']/$n, 1e6*k['accum']/$n))"
done
