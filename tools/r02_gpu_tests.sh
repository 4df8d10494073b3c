#!/bin/bash
# full GPU test suite + the bench line (N=1), outputs under gpurun_out/r02t
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r02t; mkdir -p $out
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -15 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 2500 $out/bench.json; tail -3 $out/bench.err
