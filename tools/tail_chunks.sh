#!/bin/bash
# Chunked tail items (HIBAG_TAIL_K, hibag_kernels.hip "hand-overs"): parity at sizes that use them, then the bench
# with undivided items (K = 1) and with 2 .. 8 chunks, on the HLA-B and the DRB1 shape.
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
timeout 600 python tools/parity_balanced.py 2200 3400 10000 2>&1 | tail -4
for k in 1 2 4 8; do
  HIBAG_TAIL_K=$k timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('hla-b 10000 samples, K = $k:', round(d['value']), 'samples/s', d['roofline']['kernels_ms_per_step'])"
done
for k in 1 2 4 8; do
  HIBAG_TAIL_K=$k timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --shape hla-drb1 --samples 4096 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('hla-drb1 4096 samples, K = $k:', round(d['value']), 'samples/s', d['roofline']['kernels_ms_per_step'])"
done
