#!/bin/bash
# pass 2: stored cells against evaluating the pairs again -- the model's own choice, each form forced
# (HIBAG_PASS2=stream / hybrid / recompute), and the pairs-per-cell threshold of the hybrid (HIBAG_STORE_PAIRS)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
timeout 600 python tools/parity_quick.py 2>&1 | tail -1
for mode in stream hybrid recompute; do
  HIBAG_PASS2=$mode timeout 600 python tools/parity_balanced.py 2200 3400 2>&1 | tail -1
  HIBAG_PASS2=$mode timeout 600 python tools/parity_widths.py 2>&1 | tail -1
done
for mode in auto recompute stream; do
  HIBAG_PASS2=$mode timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('hla-b 10000 samples, $mode:', round(d['value']), 'samples/s', d['roofline']['kernels_ms_per_step'])"
  HIBAG_PASS2=$mode timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --shape hla-drb1 --samples 4096 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('hla-drb1 4096 samples, $mode:', round(d['value']), 'samples/s', d['roofline']['kernels_ms_per_step'])"
done
for t in 2 4 6 8 12 16 24; do
  HIBAG_PASS2=hybrid HIBAG_STORE_PAIRS=$t timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('hla-b 10000 samples, hybrid, cells above $t pairs stored:', round(d['value']), 'samples/s', d['roofline']['kernels_ms_per_step'])"
done
