#!/bin/bash
# times tools/wide_bench.py (a model of 33..112-SNP classifiers only) with every gpurun_var_*.so at the repo root, three rounds
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for so in gpurun_var_*.so; do
  echo -n "$so  "; HIBAG_HIP_LIBRARY=$PWD/$so timeout 200 python tools/wide_bench.py < /dev/null 2>&1 | tail -1
done
done
