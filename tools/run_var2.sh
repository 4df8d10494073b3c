#!/bin/bash
# times every gpurun_var_*.so on the benchmark step, twice (variants of one call on one box are comparable); PARITY=1 first
# checks each against the oracle (ablation builds fail that on purpose)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
if [ -n "$PARITY" ]; then
for so in gpurun_var_*.so; do
  echo "== $so parity"; HIBAG_HIP_LIBRARY=$PWD/$so timeout 300 python tools/parity_quick.py 2>&1 | tail -1
done
fi
for rep in 1 2; do
for so in gpurun_var_*.so; do
  HIBAG_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$so', round(d['value']), d['roofline']['kernels_ms_per_step'])"
done
done
